"""Lane-level numpy model of mlp_bwd_fused_kernel's index algebra (csrc/mlp_fused.hip; one wave, TT = 1): dz patch -> B
fragments, dA^T = W2^T dz^T through the permuted transposing LDS read, du = dA * gelu'(u) as the A operand of GEMM 2,
dh = du W1 through the standard K-strided read.  CPU only; prints max deviations (expect ~1e-15)."""
import numpy as np

C, HC, HID = 96, 96, 384
KJ, NT, NB = C // 32, C // 16, HC // 32
rng = np.random.default_rng(1)
dz = rng.standard_normal((16, C)); W1 = rng.standard_normal((HID, C)) / 10; W2 = rng.standard_normal((C, HID)) / 10
gp = rng.standard_normal((16, HID))
ref_du = (dz @ W2) * gp
ref_dh = ref_du @ W1


def mma(c, a, b):
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for lane in range(64):
        g, r = lane >> 4, lane & 15
        A[r, g * 8:(g + 1) * 8] = a[lane]; B[g * 8:(g + 1) * 8, r] = b[lane]
    D = A @ B
    for lane in range(64):
        g, lc = lane >> 4, lane & 15
        for reg in range(4):
            c[lane][reg] += D[4 * g + reg][lc]


def frag_kc(T, r0, kk):
    return np.array([T[r0 + (lane & 15), kk + (lane >> 4) * 8: kk + (lane >> 4) * 8 + 8] for lane in range(64)])


def tr_read(T, addr):
    """ds_read_b64_tr_b16 as described in common.h: per 16-lane group, lane i supplies 4 consecutive elements at addr(i)
    = block[i>>2][4*(i&3) .. +3]; lane c receives column c of the [4][16] block.  addr: lane -> (row, col)."""
    out = np.zeros((64, 4))
    for grp in range(4):
        block = np.zeros((4, 16))
        for i in range(16):
            r, c0 = addr(grp * 16 + i)
            block[i >> 2, 4 * (i & 3): 4 * (i & 3) + 4] = T[r, c0:c0 + 4]
        for c in range(16):
            out[grp * 16 + c] = block[:, c]
    return out


def frag_ks(T, c0, klo_of, khi_of, use_tr):
    f = np.zeros((64, 8))
    if use_tr:
        f[:, :4] = tr_read(T, lambda lane: (klo_of(lane) + ((lane & 15) >> 2), c0 + ((lane & 15) & 3) * 4))
        f[:, 4:] = tr_read(T, lambda lane: (khi_of(lane) + ((lane & 15) >> 2), c0 + ((lane & 15) & 3) * 4))
    else:
        for lane in range(64):
            i = lane & 15
            for j in range(4):
                f[lane][j] = T[klo_of(lane) + j, c0 + i]; f[lane][j + 4] = T[khi_of(lane) + j, c0 + i]
    return f


def frag_ks_perm(T, h0, tsel, klo_of, khi_of, use_tr):
    f = np.zeros((64, 8))
    if use_tr:
        f[:, :4] = tr_read(T, lambda lane: (klo_of(lane) + ((lane & 15) >> 2), h0 + ((lane & 15) & 3) * 8 + tsel * 4))
        f[:, 4:] = tr_read(T, lambda lane: (khi_of(lane) + ((lane & 15) >> 2), h0 + ((lane & 15) & 3) * 8 + tsel * 4))
    else:
        for lane in range(64):
            i = lane & 15
            col = h0 + (i >> 2) * 8 + tsel * 4 + (i & 3)
            for j in range(4):
                f[lane][j] = T[klo_of(lane) + j, col]; f[lane][j + 4] = T[khi_of(lane) + j, col]
    return f


for use_tr in (True, False):
    # phase 1 tail: row layout (prow = lane>>2, q = lane&3) -> Dz patch -> fragments
    Dz = np.zeros((16, C))
    for lane in range(64):
        prow, q = lane >> 2, lane & 3
        for pp in range(KJ):
            col = pp * 32 + q * 8
            Dz[prow, col:col + 8] = dz[prow, col:col + 8]
    dzf = [frag_kc(Dz, 0, j * 32) for j in range(KJ)]
    Y = [np.zeros((64, 4)) for _ in range(NT)]
    du = np.zeros((16, HID))
    for c in range(HID // HC):
        W1c = W1[c * HC:(c + 1) * HC]            # [HC (k = hidden)][C]
        W2c = W2[:, c * HC:(c + 1) * HC]         # [C (k = channel)][HC]
        for blk in range(NB):
            U = [np.zeros((64, 4)), np.zeros((64, 4))]
            for j in range(KJ):
                for ts in range(2):
                    w = frag_ks_perm(W2c, blk * 32, ts, lambda lane: j * 32 + (lane >> 4) * 8, lambda lane: j * 32 + (lane >> 4) * 8 + 4, use_tr)
                    mma(U[ts], w, dzf[j])
            af = np.zeros((64, 8))
            for lane in range(64):
                g, lc = lane >> 4, lane & 15
                h0 = c * HC + blk * 32 + g * 8
                for ts in range(2):
                    for r in range(4):
                        af[lane][4 * ts + r] = U[ts][lane][r] * gp[lc, h0 + 4 * ts + r]
                du[lc, h0:h0 + 8] = af[lane]
            for nt in range(NT):
                w = frag_ks(W1c, nt * 16, lambda lane: blk * 32 + (lane >> 4) * 8, lambda lane: blk * 32 + (lane >> 4) * 8 + 4, use_tr)
                mma(Y[nt], af, w)
    dh = np.zeros((16, C))
    for nt in range(NT):
        for lane in range(64):
            g, lc = lane >> 4, lane & 15
            for r in range(4):
                dh[4 * g + r, nt * 16 + lc] = Y[nt][lane][r]
    print("use_tr", use_tr, "du err", np.abs(du - ref_du).max(), "dh err", np.abs(dh - ref_dh).max())
