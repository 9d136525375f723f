import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from diffusioniqt_amd import _lib
_lib.load()
B, D, H, W, Cin, Cout = 1, 4, 8, 16, 32, 64
g = torch.Generator().manual_seed(1)
x = torch.randint(-3, 4, (B, D, H, W, Cin), generator=g).float()
dy = torch.randint(-2, 3, (B, D, H, W, Cout), generator=g).float()
geo = (B, D, H, W, Cin, Cout, 3, 3, 3, 1, 1, 1, 0, 0, 0)
nb = _lib.query("diqt_conv3d_bwd_weight_h_workspace_bytes", *geo)
wr = torch.zeros(Cout, Cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
yr = F.conv3d(F.pad(x.double().permute(0, 4, 1, 2, 3), (1,) * 6), wr)
yr.backward(dy.double().permute(0, 4, 1, 2, 3))
dw = torch.empty(Cout, Cin, 3, 3, 3, device="cuda"); db = torch.empty(Cout, device="cuda"); ws = torch.empty(nb // 4, device="cuda")
_lib.call("diqt_conv3d_bwd_weight_h", x.cuda(), dy.cuda(), dw, db, ws, nb, *geo, 1, torch.cuda.current_stream().cuda_stream)
bad = (dw.cpu().double() != wr.grad)
print("mismatch fraction", bad.float().mean().item())
print("by co block of 8:", [round(bad[c:c + 8].float().mean().item(), 2) for c in range(0, Cout, 8)])
print("by ci block of 4:", [round(bad[:, c:c + 4].float().mean().item(), 2) for c in range(0, Cin, 4)])
print("by tap:", [round(bad.reshape(Cout, Cin, 27)[:, :, t].float().mean().item(), 2) for t in range(27)])
print("db ok:", torch.equal(db.cpu().double(), dy.double().sum(dim=(0, 1, 2, 3))))
ref_db = dy.double().sum(dim=(0, 1, 2, 3))
print("db", db.cpu()[:8].tolist(), "ref", ref_db[:8].tolist())
print("dw[0,0]", dw.cpu()[0, 0].flatten().tolist())
print("ref    ", wr.grad[0, 0].flatten().tolist())
print("dw[40,17]", dw.cpu()[40, 17].flatten().tolist())
print("ref      ", wr.grad[40, 17].flatten().tolist())
# per-slice slabs: which tiles contribute
T = 27
ks = 4
sl = ws[:ks * Cout * Cin * T].reshape(ks, Cout, Cin, T).cpu().double()
for k in range(ks):
    # reference contribution of tile k: tiles enumerate (tz, ty, tx) with tx fastest; tile = 2 x 4 x 16
    tz, ty = divmod(k, 2)
    dyk = torch.zeros_like(dy); dyk[:, 2 * tz:2 * tz + 2, 4 * ty:4 * ty + 4] = dy[:, 2 * tz:2 * tz + 2, 4 * ty:4 * ty + 4]
    w2 = torch.zeros(Cout, Cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(F.pad(x.double().permute(0, 4, 1, 2, 3), (1,) * 6), w2).backward(dyk.double().permute(0, 4, 1, 2, 3))
    print("slice", k, "mismatch", (sl[k].reshape(Cout, Cin, 3, 3, 3) != w2.grad).float().mean().item())
