#!/usr/bin/env python3
"""CPU model of the LDS images of ppo_grad_split_kernel (dril.jl_amd/csrc/dril_kernels.hip): checks, with exact integer data and the
documented lane maps (cdna_hip_programming.md §3 "A/B operand lane maps", T10 ds_read_b64_tr_b16; MI355X_MICROARCH.md §LDS banking), that

  1. the ONE swizzled weight image gives W2 . X through row reads (L2 forward) and W2' . Z through transposed reads (dh1),
  2. the per-wave transposed activation images give dW2 = Z . X' (contraction over the tile's 32 samples),
  3. every one of those accesses is bank-conflict-free (row reads / transposed reads: two 32-lane halves on 64 banks; 8-byte stores:
     four 16-lane groups on 32 banks).

The address arithmetic below is a transcription of the kernel's (same names); run it after touching either.  No GPU needed.
"""
import numpy as np

H = 64


def rowfn(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def gw(r):
    return (((r >> 1) & 1) << 3) | (((r >> 2) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 4) & 1)


def gs(r):
    return (((r >> 1) & 1) << 3) | (((r >> 2) & 1) << 2) | (((r >> 3) & 1) << 1) | (r & 1)


def mfma(Afrag, Bfrag):
    """v_mfma_f32_32x32x16: Afrag/Bfrag [64 lanes][8]; -> D[32][32] in matrix form"""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for l in range(64):
        for j in range(8):
            A[l & 31, 8 * (l >> 5) + j] = Afrag[l][j]
            B[8 * (l >> 5) + j, l & 31] = Bfrag[l][j]
    return A @ B


def to_acc(D):
    """matrix [32][32] -> accumulator layout [64 lanes][16 regs]"""
    acc = np.zeros((64, 16))
    for l in range(64):
        for r in range(16):
            acc[l, r] = D[rowfn(r, l >> 5), l & 31]
    return acc


def conflicts_read64(addrs):
    """ds_read_b64 / ds_read_b64_tr_b16: halves of 32 lanes, bank = (a / 4) % 64, each lane touches two banks; -> extra cycles"""
    extra = 0
    for half in range(2):
        use = {}
        for l in range(32 * half, 32 * half + 32):
            for d in (0, 4):
                use.setdefault(((addrs[l] + d) // 4) % 64, set()).add((addrs[l] + d) // 4)
        extra += max(len(v) for v in use.values()) - 1
    return extra


def conflicts_write64(addrs):
    """ds_write_b64: four groups of 16 contiguous lanes, bank = (a / 4) % 32"""
    extra = 0
    for g in range(4):
        use = {}
        for l in range(16 * g, 16 * g + 16):
            for d in (0, 4):
                use.setdefault(((addrs[l] + d) // 4) % 32, set()).add((addrs[l] + d) // 4)
        extra += max(len(v) for v in use.values()) - 1
    return extra


def tr_read(mem, addrs):
    """ds_read_b64_tr_b16 on a byte-addressed array of 16-bit elements `mem` (indexed by byte // 2): per 16-lane group, lane 4q+p supplies
    the address of row q, columns 4p..4p+3; lane i receives column i, row q in element q"""
    out = np.zeros((64, 4))
    for g in range(4):
        for i in range(16):
            for q in range(4):
                src = addrs[16 * g + 4 * q + (i >> 2)]
                out[16 * g + i, q] = mem[src // 2 + (i & 3)]
    return out


def main():
    rng = np.random.default_rng(0)
    W2 = rng.integers(-8, 8, (H, H)).astype(float)          # [out = h2 unit][in = h1 unit]
    X = rng.integers(-8, 8, (H, 32)).astype(float)          # h1 [unit][sample]
    Z = rng.integers(-8, 8, (H, 32)).astype(float)          # dz2 [unit][sample]
    total_conf = 0

    # ---- weight image (one piece): row o, pair kp -> byte o*128 + (((kp>>1) ^ gw(o)) << 3) + ((kp&1) << 2) --------------------------------
    img = np.zeros(H * 128 // 2)
    for o in range(H):
        for kp in range(H // 2):
            byte = o * 128 + ((((kp >> 1) ^ gw(o)) & 15) << 3) + ((kp & 1) << 2)
            img[byte // 2], img[byte // 2 + 1] = W2[o, 2 * kp], W2[o, 2 * kp + 1]

    accX = [to_acc(X[32 * m:32 * m + 32]) for m in range(2)]     # activations in accumulator layout
    accZ = [to_acc(Z[32 * m:32 * m + 32]) for m in range(2)]
    # L2 forward: Y = W2 . X
    Y = np.zeros((H, 32))
    for mo in range(2):
        D = np.zeros((32, 32))
        for mi in range(2):
            for s in range(2):
                Af, Bf, a0s, a1s = np.zeros((64, 8)), np.zeros((64, 8)), [], []
                for l in range(64):
                    c, h = l & 31, l >> 5
                    wf_base = c * 128 + (((h ^ gw(c)) & 15) << 3)
                    a0 = (wf_base ^ (64 * mi + 32 * s)) + 4096 * mo
                    a1 = a0 ^ 16
                    a0s.append(a0); a1s.append(a1)
                    Af[l, 0:4] = img[a0 // 2:a0 // 2 + 4]; Af[l, 4:8] = img[a1 // 2:a1 // 2 + 4]
                    Bf[l] = accX[mi][l, 8 * s:8 * s + 8]
                total_conf += conflicts_read64(a0s) + conflicts_read64(a1s)
                D += mfma(Af, Bf)
        Y[32 * mo:32 * mo + 32] = D
    assert np.array_equal(Y, W2 @ X), "L2 forward (row reads of the weight image)"
    # dh1: G = W2' . Z
    G = np.zeros((H, 32))
    for mk in range(2):
        D = np.zeros((32, 32))
        for mi in range(2):
            for s in range(2):
                a0s, a1s = [], []
                for l in range(64):
                    h, tg, e = l >> 5, (l >> 4) & 1, l & 15
                    tq, tp = e >> 2, e & 3
                    wt_base = (4 * h + tq) * 128 + ((((4 * tg + tp) ^ (8 * (tq >> 1) + 4 * h)) & 15) << 3)
                    a0 = (wt_base ^ (64 * mk + 2048 * s + 8 * s)) + 4096 * mi
                    a0s.append(a0); a1s.append(a0 ^ (1024 | 16))
                total_conf += conflicts_read64(a0s) + conflicts_read64(a1s)
                Af = np.concatenate([tr_read(img, a0s), tr_read(img, a1s)], axis=1)
                Bf = np.stack([accZ[mi][l, 8 * s:8 * s + 8] for l in range(64)])
                D += mfma(Af, Bf)
        G[32 * mk:32 * mk + 32] = D
    assert np.array_equal(G, W2.T @ Z), "dh1 (transposed reads of the weight image)"

    # ---- transposed activation images: store_pieces_T / load_frag_T ---------------------------------------------------------------------------
    def store_T(acc):
        T = np.zeros(32 * 128 // 2)
        for m in range(2):
            for g in range(4):
                addrs = []
                for l in range(64):
                    c, h = l & 31, l >> 5
                    base = c * 128 + (((h ^ gs(c)) & 15) << 3)
                    a = base ^ (64 * m + 16 * g)
                    addrs.append(a)
                    T[a // 2:a // 2 + 4] = acc[m][l, 4 * g:4 * g + 4]
                nonlocal total_conf
                total_conf += conflicts_write64(addrs)
        return T

    def load_T(T, m, s):
        a0s, a1s = [], []
        for l in range(64):
            h, gm, e = l >> 5, (l >> 4) & 1, l & 15
            q, p = e >> 2, e & 3
            rbase = (8 * h + q) * 128 + ((((4 * gm + p) ^ (8 * (q >> 1) + 2 * h + (q & 1))) & 15) << 3)
            a = (rbase ^ (64 * m)) + 2048 * s
            a0s.append(a); a1s.append(a ^ (512 | 32))
        nonlocal total_conf
        total_conf += conflicts_read64(a0s) + conflicts_read64(a1s)
        return np.concatenate([tr_read(T, a0s), tr_read(T, a1s)], axis=1)
    TX, TZ = store_T(accX), store_T(accZ)
    dW2 = np.zeros((H, H))
    for mi in range(2):
        for mj in range(2):
            D = np.zeros((32, 32))
            for s in range(2):
                D += mfma(load_T(TZ, mi, s), load_T(TX, mj, s))
            dW2[32 * mi:32 * mi + 32, 32 * mj:32 * mj + 32] = D
    assert np.array_equal(dW2, Z @ X.T), "dW2 (transposed activation images)"
    assert total_conf == 0, f"{total_conf} extra LDS cycles from bank conflicts"
    print("weight image: L2 forward and dh1 exact; activation images: dW2 exact; all reads / 8-byte stores bank-conflict-free")


if __name__ == "__main__":
    main()
