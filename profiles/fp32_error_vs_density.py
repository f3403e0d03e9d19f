"""fp32 pipeline error vs the nominee margin at extreme activity densities (large DC term)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from ffsubsync_amd import _native
def direct(ref_pm, s_pm, N):
    a = np.zeros(N); a[:len(s_pm)] = s_pm
    b = np.zeros(N); b[:len(ref_pm)] = ref_pm
    return np.real(np.fft.ifft(np.conj(np.fft.fft(a)) * np.fft.fft(b)))
for log2n, R, S in [(20, 720000, 750000), (21, 720000, 750000), (21, 1000000, 1090000)]:
    N = 1 << log2n
    if R + S > N and log2n == 20:
        pass  # circular wrap is fine for an error measurement
    plan = _native.Plan(N, 1, 2)
    for dens in (0.02, 0.2, 0.5, 0.98):
        rng = np.random.RandomState(int(dens * 100))
        ref = (rng.rand(R) < dens).astype(np.uint8); a = (rng.rand(S) < dens).astype(np.uint8); b = (rng.rand(S) < 1 - dens).astype(np.uint8)
        d = lambda x: torch.from_numpy(x).cuda()
        oa, ob = plan.correlate_full(_native.FFS_DTYPE_U8, d(ref), (0, 1), d(a), (0, 1), d(b), (0, 1))
        ea = direct(2.0 * ref - 1, 2.0 * a - 1, N); eb = direct(2.0 * ref - 1, 2.0 * b - 1, N)
        margin = 0.5 * 5.96e-8 * log2n * np.sqrt(R * S)
        print("N=2^%d density %.2f: max err a %.4f b %.4f  margin %.3f  ratio %.1f" % (log2n, dens, np.abs(oa.cpu().numpy() - ea).max(), np.abs(ob.cpu().numpy() - eb).max(), margin, margin / max(np.abs(oa.cpu().numpy() - ea).max(), np.abs(ob.cpu().numpy() - eb).max())))
    plan.close()
