"""Lane-level numpy model of csrc/mlp_fused.hip's index algebra (one wave, TT = 1): W1 rows permuted into LDS, U^T accumulators reused as
the A operand of GEMM 2, under the library's MFMA fragment convention (common.h).  CPU only; prints the max deviation from
gelu(h W1^T + b1) W2^T (expect ~1e-15)."""
import numpy as np
from math import erf
C, HC, HID = 96, 96, 384
KJ, NT, NB = C//32, C//16, HC//32
rng = np.random.default_rng(0)
h = rng.standard_normal((16, C)); W1 = rng.standard_normal((HID, C))/10; W2 = rng.standard_normal((C, HID))/10; b1 = rng.standard_normal(HID)
gelu = np.vectorize(lambda x: 0.5*x*(1+erf(x/np.sqrt(2))))
ref_act = gelu(h@W1.T + b1); ref = ref_act@W2.T
def mma(c, a, b):  # a,b: [64][8] lane fragments, c: [64][4]
    A = np.zeros((16,32)); B = np.zeros((32,16))
    for lane in range(64):
        g, r = lane>>4, lane&15
        A[r, g*8:(g+1)*8] = a[lane]; B[g*8:(g+1)*8, r] = b[lane]
    D = A@B
    for lane in range(64):
        g, lc = lane>>4, lane&15
        for reg in range(4): c[lane][reg] += D[4*g+reg][lc]
def frag_kc(T, r0, kk):
    return np.array([T[r0+(lane&15), kk+(lane>>4)*8: kk+(lane>>4)*8+8] for lane in range(64)])
hf = [np.array([h[lane&15, j*32+(lane>>4)*8: j*32+(lane>>4)*8+8] for lane in range(64)]) for j in range(KJ)]
Y = [np.zeros((64,4)) for _ in range(NT)]
act = np.zeros((16, HID))
for c in range(HID//HC):
    W1c = np.zeros((HC, C)); W2c = np.zeros((C, HC)); b1c = b1[c*HC:(c+1)*HC]
    for i in range(HC*C//8):
        x, k8 = i//(C//8), (i%(C//8))*8
        y = x & 31
        rho = (x & ~31) + (((y>>2)&1)<<4) + ((y>>3)<<2) + (y&3)
        W1c[rho, k8:k8+8] = W1[c*HC:(c+1)*HC].reshape(-1)[i*8:i*8+8]
    for i in range(C*HC//8):
        row, c8 = i//(HC//8), (i%(HC//8))*8
        W2c[row, c8:c8+8] = W2[row, c*HC+c8: c*HC+c8+8]
    for blk in range(NB):
        U = [np.zeros((64,4)), np.zeros((64,4))]
        for j in range(KJ):
            for t in range(2):
                mma(U[t], frag_kc(W1c, blk*32+t*16, j*32), hf[j])
        af = np.zeros((64,8))
        for lane in range(64):
            g, lc = lane>>4, lane&15
            bias = b1c[blk*32+g*8: blk*32+g*8+8]
            for t in range(2):
                for r in range(4):
                    af[lane][4*t+r] = gelu(U[t][lane][r] + bias[4*t+r])
            act[lc, c*HC+blk*32+g*8: c*HC+blk*32+g*8+8] = af[lane]
        for nt in range(NT):
            mma(Y[nt], af, frag_kc(W2c, nt*16, blk*32))
out = np.zeros((16, C))
for nt in range(NT):
    for lane in range(64):
        g, lc = lane>>4, lane&15
        for r in range(4): out[4*g+r, nt*16+lc] = Y[nt][lane][r]
print("act err", np.abs(act-ref_act).max(), "out err", np.abs(out-ref).max())
