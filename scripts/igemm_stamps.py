"""In-kernel wall-clock stamps of the 16-wave tiles.  Needs a diagnostic build of the library:
`make -C lbm_amd/csrc clean && make -C lbm_amd/csrc DIAG=1 -j8` (the product build has neither tile ids 115 / 116 nor
lbmk_debug_igemm_times)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lbm_amd import ops, _C

dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
fn = _C.lib.lbmk_debug_igemm_times
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int]
for (tile, bm, bn, B, H, cin, cout) in [(115, 512, 128, 4, 1024, 128, 128), (115, 512, 128, 4, 1024, 256, 128),
                                         (116, 256, 256, 16, 256, 512, 512), (117, 256, 256, 16, 256, 512, 512)]:
    x = torch.randn(B, H, H, cin, device=dev, generator=g).bfloat16()
    w = torch.randn(cout, cin, 3, 3, device=dev, generator=g).bfloat16() * (9 * cin) ** -0.5
    pw = ops.pack_conv_weight(w, torch.zeros(cout, device=dev))
    for _ in range(int(os.environ.get('STAMP_ITERS', '3'))):
        ops.conv2d(x, pw, tile=tile)
    torch.cuda.synchronize()
    nb = min(65536, (B * H * H // bm) * ((cout + bn - 1) // bn))
    buf = np.zeros((nb, 8), dtype=np.uint64)
    assert fn(buf.ctypes.data, nb) == 0
    t = buf.astype(np.int64)
    t6 = t
    clk = (t[:, 7] - t[:, 6]) / np.maximum(t[:, 3] - t[:, 2], 1) * 0.1   # GHz
    t = t[:, :6]
    seg = np.diff(t, axis=1) * 0.01   # us
    print(f"tile {tile} conv {cin}->{cout} {H}^2 B{B}: blocks {nb}")
    print("  median us: setup %.2f  first-stage %.2f  mainloop %.2f  epilogue-issue %.2f  store-drain %.2f  total %.2f" % (
        *np.median(seg, axis=0), np.median(t[:, 5] - t[:, 0]) * 0.01))
    cyc = np.median(t6[:, 7] - t6[:, 6]) / (9 * cin // 64)
    print("  shader cycles per 64-deep K step: %.0f" % cyc)
    print("  in-kernel clock over the K loop: median %.2f GHz (p10 %.2f, p90 %.2f)" % (np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
    span = (t[:, 5].max() - t[:, 0].min()) * 0.01
    print("  kernel span %.1f us; sum(block time)/256 = %.1f us; rounds %.1f" % (span, (t[:, 5] - t[:, 0]).sum() * 0.01 / 256, nb / 256))
    # turnover: sort starts; the k-th start beyond the first 256 pairs with the k-th end
    starts = np.sort(t[:, 0]); ends = np.sort(t[:, 5])
    if nb > 512:
        gap = (starts[256:] - ends[:nb - 256]) * 0.01
        print("  turnover (start of block 256+k minus k-th end): median %.2f us, p90 %.2f" % (np.median(gap), np.percentile(gap, 90)))
