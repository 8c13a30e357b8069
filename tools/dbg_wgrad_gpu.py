import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morefusion_amd import _lib
L = _lib.lib(); p = lambda t: t.data_ptr(); st = _lib.stream_ptr
def run(dY, A):
    M, N = dY.shape; K = A.shape[1]
    dW = torch.zeros(N, K, device="cuda"); ws = torch.zeros(N * K, device="cuda")
    _lib.check(L.mf_linear_wgrad_bf16(p(dY), 0, N, p(A), 0, K, p(dW), 0, K, p(ws), M, N, K, 1, 1, st()), "wgrad")
    torch.cuda.synchronize()
    return dW.cpu()
bf = torch.bfloat16
M, N, K = 64, 128, 128
m_idx = torch.arange(M)[:, None]; j_idx = torch.arange(K)[None, :]
code = ((m_idx * 2 + j_idx * 3) % 251 + 1).float()          # Q[m][j], exact in bf16 (<= 251 < 256)
Q = code.to(bf).cuda()
P = torch.zeros(M, N, dtype=bf, device="cuda"); P[torch.arange(M), torch.arange(M)] = 1   # i < 64 selects m = i
for rep in range(3):
    w = run(P, Q)                       # w[i][j] = Q[i][j] for i < 64
    got = w[:64]
    bad = (got != code).nonzero()
    print("Q-image rep", rep, "bad", len(bad))
    for b in bad[:24].tolist():
        m, j = b
        g = float(got[m, j]); src = (code == g).nonzero().tolist()
        near = [s for s in src if abs(s[1] - j) <= 3][:4]
        print("   [m=%d, j=%d] want %g got %g ; same value lives at (m,j) near: %s" % (m, j, float(code[m, j]), g, near))
# P image: swap roles (wgrad of transposed problem): dW[i][j] = sum_m P[m][i] Q[m][j]; make Q one-hot to read P
P2 = code[:, :N].to(bf).cuda()          # P[m][i] codes
Q2 = torch.zeros(M, K, dtype=bf, device="cuda"); Q2[torch.arange(M), torch.arange(M)] = 1   # j < 64 selects m = j
for rep in range(2):
    w = run(P2, Q2)                     # w[i][j] = P[j][i] for j < 64
    got = w[:, :64].t()                 # [m][i]
    bad = (got != code[:, :N]).nonzero()
    print("P-image rep", rep, "bad", len(bad))
    for b in bad[:24].tolist():
        m, i = b
        g = float(got[m, i]); src = (code == g).nonzero().tolist()
        near = [s for s in src if abs(s[1] - i) <= 3][:4]
        print("   [m=%d, i=%d] want %g got %g ; near: %s" % (m, i, float(code[m, i]), g, near))
