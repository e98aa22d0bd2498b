import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
from spiral_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
n, i, h = 20000, 128, 128
for kind, g in (('gru', 3), ('lstm', 4), ('rnn', 1)):
    x = torch.randn(n, i, device=dev) * 2
    w_ih = torch.randn(g*h, i, device=dev) * 0.2
    w_hh = torch.randn(g*h, h, device=dev) * 0.2
    b_ih = torch.randn(g*h, device=dev) * 0.1; b_hh = torch.randn(g*h, device=dev) * 0.1
    mem0 = torch.randn(n, h, device=dev); c0 = torch.randn(n, h, device=dev)
    outs = {}
    for mode in ('fp32', 'bf16x3', 'f16x3'):
        ops.CELL_GEMM = mode
        mh, mc = mem0.clone(), c0.clone()
        with torch.no_grad():
            if kind == 'gru': o = ops.gru_cell(x, w_ih, w_hh, b_ih, b_hh, mh)
            elif kind == 'lstm': o = ops.lstm_cell(x, w_ih, w_hh, b_ih, b_hh, mh, mc); o = o[0] if isinstance(o, tuple) else o
            else: o = ops.rnn_cell(x, w_ih, w_hh, b_ih, b_hh, mh)
        outs[mode] = o.double().cpu()
    # float64 reference
    xd, hd = x.double().cpu(), mem0.double().cpu()
    gi = xd @ w_ih.double().cpu().t() + b_ih.double().cpu(); gh = hd @ w_hh.double().cpu().t() + b_hh.double().cpu()
    if kind == 'gru':
        r = torch.sigmoid(gi[:, :h] + gh[:, :h]); z = torch.sigmoid(gi[:, h:2*h] + gh[:, h:2*h])
        nn_ = torch.tanh(gi[:, 2*h:] + r * gh[:, 2*h:]); ref = (1 - z) * nn_ + z * hd
    elif kind == 'lstm':
        s = gi + gh; ig, fg, gg, og = torch.sigmoid(s[:, :h]), torch.sigmoid(s[:, h:2*h]), torch.tanh(s[:, 2*h:3*h]), torch.sigmoid(s[:, 3*h:])
        cy = fg * c0.double().cpu() + ig * gg; ref = og * torch.tanh(cy)
    else:
        ref = torch.tanh(gi + gh)
    for mode in outs:
        print(kind, mode, 'max abs err vs float64:', float((outs[mode] - ref).abs().max()))
