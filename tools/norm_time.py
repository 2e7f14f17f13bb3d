"""time vlr_rmsnorm_fwd_f32 / vlr_rmsnorm_fwd at the step's shape and print a checksum of the outputs:
   VLR_NORM_FWD_REG=0 python tools/norm_time.py ; VLR_NORM_FWD_REG=1 python tools/norm_time.py   (same checksum: the two forms are bit-identical)"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vl-rlhf_amd"))
from vlrlhf import _hip  # noqa: E402

M, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12792, 4096)
g = torch.Generator().manual_seed(0)
x32 = torch.randn(M, H, generator=g).cuda()
x16 = x32.bfloat16()
w = (1 + 0.1 * torch.randn(H, generator=g)).bfloat16().cuda()
y = torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
rstd = torch.empty(M, dtype=torch.float32, device="cuda")
for name, x in (("vlr_rmsnorm_fwd_f32", x32), ("vlr_rmsnorm_fwd", x16)):
    for _ in range(5):
        _hip.call(name, x, w, y, rstd, M, H, 1e-5)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200):
        _hip.call(name, x, w, y, rstd, M, H, 1e-5)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / 200
    by = M * H * (x.element_size() + 2)
    h = hashlib.sha1(y.cpu().view(torch.int16).numpy().tobytes() + rstd.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"{name} [{M} x {H}] reg={os.environ.get('VLR_NORM_FWD_REG', '1')}: {us:.1f} us, {by / us / 1e6:.2f} TB/s, checksum {h}")

# backward on the fp32 stream: dx = rstd (dy w - xhat mean(dy w xhat)) + dres, dw partials reduced in two deterministic stages
dy = torch.randn(M, H, generator=g).bfloat16().cuda()
dres = torch.randn(M, H, generator=g).bfloat16().cuda()
dx = torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
dw = torch.zeros(H, dtype=torch.bfloat16, device="cuda")
ws = torch.empty(_hip.helper("vlr_rmsnorm_bwd_workspace_bytes", H), dtype=torch.uint8, device="cuda")
_hip.call("vlr_rmsnorm_fwd_f32", x32, w, y, rstd, M, H, 1e-5)
for _ in range(5):
    _hip.call("vlr_rmsnorm_bwd_f32", dy, x32, w, rstd, dres, dx, dw, 0, ws, M, H)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(200):
    _hip.call("vlr_rmsnorm_bwd_f32", dy, x32, w, rstd, dres, dx, dw, 0, ws, M, H)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) * 1000 / 200
h = hashlib.sha1(dx.cpu().view(torch.int16).numpy().tobytes() + dw.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
print(f"vlr_rmsnorm_bwd_f32 (+ dw reduction) [{M} x {H}] early={os.environ.get('VLR_NORM_BWD_EARLY', '1')}: {us:.1f} us, {M * H * 10 / us / 1e6:.2f} TB/s, checksum {h}")
