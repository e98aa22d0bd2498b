"""Prints the per-step state error against float64 of a 2-layer GRU run for 32 snapshots in
each arithmetic (see tests/test_gpu_drift.py), plus torch's own fp32 GRU on the GPU and CPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spiral_amd import ops
from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder.implicitTimeModels import GRU
from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP
from tests.test_gpu_drift import _run, STEPS

dev = torch.device("cuda:0")
n, h = 4096, 128
spread = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
torch.manual_seed(3)
dec = GRU(h, h, n, 2, MLP(h, [8])).to(dev)
dec.eval()
start = torch.randn(2, n, h, device=dev)
gen = torch.Generator().manual_seed(5)
xs = [(torch.randn(n, h, generator=gen) * torch.exp(spread * torch.randn(n, 1, generator=gen))).to(dev) for _ in range(STEPS)]
ref = torch.nn.GRU(h, h, num_layers=2, batch_first=True).double()
ref.load_state_dict({k: v.double().cpu() for k, v in dec.base_model.state_dict().items()})
state = start.double().cpu(); want = []
with torch.no_grad():
    for x in xs:
        _, state = ref(x.double().cpu()[:, None, :], state.contiguous()); want.append(state.clone())
def show(name, got):
    errs = [float((g - w).abs().max()) for g, w in zip(got, want)]
    print(f"{name:20s} abs " + " ".join(f"{e:.1e}" for e in errs[::4] + errs[-1:]), flush=True)
for name, gemm, rows in (("f16x3 whole-table", "f16x3", True), ("f16x3 row-tile", "f16x3", False), ("bf16x3", "bf16x3", False), ("fp32 mfma", "fp32", False)):
    show(name, _run(dec, xs, start, gemm, rows))
for d in ("cuda", "cpu"):
    g32 = torch.nn.GRU(h, h, num_layers=2, batch_first=True).to(d)
    g32.load_state_dict(dec.base_model.state_dict())
    st = start.to(d).clone(); got = []
    with torch.no_grad():
        for x in xs:
            _, st = g32(x.to(d)[:, None, :], st.contiguous()); got.append(st.double().cpu())
    show(f"torch fp32 {d}", got)
print("max |state|", [round(float(w.abs().max()), 2) for w in want[::8]])
