"""spiral_amd: MI355X-native implementation of the spiro.dtdg per-snapshot GCN
message-passing path (degree/norm, COO->CSR snapshot build, fused normalised aggregation,
GRU/LSTM/RNN node-memory update) behind the reference's own layer API.

Import paths mirror the reference package `spiro` for the modules on that path, e.g.
`spiro.dtdg.types` -> `spiral_amd.dtdg.types`.  All compute is in libspmi (HIP, gfx950);
there is no CPU implementation.
"""
__version__ = "0.1.0"
