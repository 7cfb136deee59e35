"""Accuracy of the matrix engines on a [2048, 256] x [256, 256] product of SiLU activations and uniform weights, emulated on the CPU (numpy; one fp32
rounding of the accumulator per MFMA, the products inside an MFMA exact): the fp32 fmaf chain / BLAS sgemm the reference runs, the three-piece bf16
split (six MFMAs per product) and the half engine (two fp16 pieces, three MFMAs; weights times 1024) - each against the float64 product.
DESIGN.md section 4a-v quotes these numbers."""
import numpy as np

rng = np.random.default_rng(0)
M, K, N = 2048, 256, 256
pre = rng.normal(0, 1.5, size=(M, K)).astype(np.float32)
A = (pre / (1 + np.exp(-pre))).astype(np.float32)
W = rng.uniform(-1 / 16, 1 / 16, size=(K, N)).astype(np.float32)
exact = A.astype(np.float64) @ W.astype(np.float64)


def bf16_round(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def split(x, n, to):
    out, r = [], x.astype(np.float32)
    for _ in range(n):
        p = to(r)
        out.append(p)
        r = (r - p).astype(np.float32)
    return out


def mfma_sum(a_pieces, w_pieces, pairs):
    acc = np.zeros((M, N), np.float32)
    for kb in range(0, K, 16):
        s = slice(kb, kb + 16)
        for i, j in pairs:
            acc = (acc.astype(np.float64) + a_pieces[i][:, s].astype(np.float64) @ w_pieces[j][s].astype(np.float64)).astype(np.float32)
    return acc


f16 = lambda r: r.astype(np.float16).astype(np.float32)
r_bf = mfma_sum(split(A, 3, bf16_round), split(W, 3, bf16_round), [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)])
SC = 1024.0
r_h = (mfma_sum(split(A, 2, f16), split(W * SC, 2, f16), [(1, 0), (0, 1), (0, 0)]) / SC).astype(np.float32)
acc = np.zeros((M, N), np.float32)
for k in range(K):
    acc = (acc.astype(np.float64) + A[:, k:k + 1].astype(np.float64) * W[k:k + 1].astype(np.float64)).astype(np.float32)
for name, r in (('fp32 fmaf chain', acc), ('BLAS sgemm', A @ W), ('bf16 x 3, six products', r_bf), ('fp16 x 2, three products (half engine)', r_h)):
    e = r.astype(np.float64) - exact
    print(f'{name:42s} rms error {np.sqrt((e ** 2).mean()):.3e}   max {np.abs(e).max():.3e}   (rms of the result {np.sqrt((exact ** 2).mean()):.3f})')
