"""TEST INFRASTRUCTURE — a CPU interpreter of the mask-tail op table (medplib_amd/tail_program.py packs it, csrc/tail_program.hip runs it).

Only tests/ may import this module.  It executes the SAME packed table the HIP kernel executes — phase by phase, op by op, with numpy in
fp32 — on operands that live in host memory (the builder only records addresses, so a program built over CPU tensors addresses host
memory).  Two uses: (1) the lowering itself — phases, hazards, the generated backward — is checked on the CPU against torch autograd of the
oracle's mask decoder (oracle/sam.py: transformer.py:62-106,151-244, mask_decoder.py:113-153) without a GPU; (2) on the GPU every op type
is checked against this interpreter on the same table.  Per-op semantics follow the kernel's comments one to one; sums run in numpy's
order, not the kernel's (tolerances, not bit-equality, against the device)."""
import ctypes

import numpy as np

from medplib_amd import tail_program as TP


def _mat(slots, addr, rows, cols, ld, cs=1):
    """A writable fp32 [rows, cols] view (row stride ld floats, column stride cs floats) of host memory at the operand address."""
    addr = int(addr)
    if addr == 0:
        return None
    base = int(slots[addr >> 56]) + (addr & ((1 << 56) - 1))
    if rows <= 0 or cols <= 0:
        return np.zeros((max(rows, 0), max(cols, 0)), np.float32)
    cs = max(int(cs), 1)
    count = (rows - 1) * ld + (cols - 1) * cs + 1
    flat = np.ctypeslib.as_array((ctypes.c_float * count).from_address(base))
    return np.lib.stride_tricks.as_strided(flat, shape=(rows, cols), strides=(4 * ld, 4 * cs))


def _mat_bf16(slots, addr, rows, cols, ld):
    """bf16 values (row stride ld ELEMENTS) read as fp32."""
    addr = int(addr)
    base = int(slots[addr >> 56]) + (addr & ((1 << 56) - 1))
    count = (rows - 1) * ld + cols
    flat = np.ctypeslib.as_array((ctypes.c_uint16 * count).from_address(base))
    u = np.lib.stride_tricks.as_strided(flat, shape=(rows, cols), strides=(2 * ld, 2)).astype(np.uint32) << 16
    return u.view(np.float32)


def _wrap_rows(x, rows_total, period):
    idx = np.arange(rows_total)
    return x[idx % period] if period > 0 else x[idx]


def run_op(op, slots):
    t, fl = int(op["type"]), int(op["flags"])
    M, N, K = int(op["M"]), int(op["N"]), int(op["K"])
    ld, p = [int(v) for v in op["ld"]], [int(v) for v in op["p"]]
    f0 = np.float32(op["f0"])
    if t == TP.OP_GEMM:
        ta, tb = bool(fl & TP.F_TRANS_A), bool(fl & TP.F_TRANS_B)
        splits = max(int(op["i0"]), 1)
        A = (_mat_bf16 if fl & TP.F_A_BF16 else _mat)(slots, p[0], K if ta else M, M if ta else K, ld[0]).copy()
        B = _mat(slots, p[1], N if tb else K, K if tb else N, ld[1]).copy()
        if p[4]:
            per = int(op["i1"])
            A2 = _mat(slots, p[4], per if per > 0 else A.shape[0], A.shape[1], ld[6])
            A = A + _wrap_rows(A2, A.shape[0], per)
        if p[5]:
            per = int(op["i2"])
            B2 = _mat(slots, p[5], per if per > 0 else B.shape[0], B.shape[1], ld[7])
            B = B + _wrap_rows(B2, B.shape[0], per)
        opA = A.T if ta else A
        opB = B.T if tb else B
        if splits > 1:
            per = -(-(-(-K // splits)) // 64) * 64
            for sp in range(splits):
                k0, k1 = sp * per, min(K, (sp + 1) * per)
                C = _mat(slots, p[2] + 4 * sp * ld[8], M, N, ld[2])
                C[...] = opA[:, k0:k1].astype(np.float32) @ opB[k0:k1].astype(np.float32)
                if p[9]:
                    _mat(slots, p[9] + 4 * sp * M, 1, M, M)[0] = A[k0:k1].sum(0)
            return
        if p[9]:
            cs = _mat(slots, p[9], 1, M, M)
            s = A.sum(0)
            cs[0] = cs[0] + s if fl & TP.F_CS_ACCUM else s
        v = f0 * (opA @ opB)
        if p[3]:
            v = v + _mat(slots, p[3], 1, N, N)
        if p[6]:
            v = v + _mat(slots, p[6], M, N, ld[3])
        if fl & TP.F_RELU:
            v = np.maximum(v, 0)
        if p[7]:
            v = np.where(_mat(slots, p[7], M, N, ld[4]) > 0, v, 0)
        v = v.astype(np.float32)
        if p[8]:
            C2 = _mat(slots, p[8], M, N, ld[5])
            C2[...] = C2 + v
        C = _mat(slots, p[2], M, N, ld[2], ld[9])
        C[...] = C + v if fl & TP.F_ACCUM else v
    elif t == TP.OP_REDUCE:
        acc = np.zeros((M, N), np.float32)
        for s in range(K):
            acc = acc + _mat(slots, p[0] + 4 * s * ld[2], M, N, ld[0])
        v = f0 * acc
        if p[2]:
            v = v + _mat(slots, p[2], 1, N, N)
        if p[3]:
            v = v + _mat(slots, p[3], M, N, ld[3])
        if fl & TP.F_RELU:
            v = np.maximum(v, 0)
        if p[4]:
            v = np.where(_mat(slots, p[4], M, N, ld[4]) > 0, v, 0)
        out = _mat(slots, p[1], M, N, ld[1], ld[5])
        out[...] = (out + v if fl & TP.F_ACCUM else v).astype(np.float32)
    elif t == TP.OP_LN_FWD:
        x = _mat(slots, p[0], M, N, ld[0])
        w, b = _mat(slots, p[1], 1, N, N)[0], _mat(slots, p[2], 1, N, N)[0]
        mean = x.mean(1, dtype=np.float32)
        var = ((x - mean[:, None]) ** 2).mean(1, dtype=np.float32)
        rstd = (1.0 / np.sqrt(var + f0)).astype(np.float32)
        _mat(slots, p[3], M, N, ld[1])[...] = (x - mean[:, None]) * rstd[:, None] * w + b
        _mat(slots, p[4], 1, M, M)[0] = mean
        _mat(slots, p[5], 1, M, M)[0] = rstd
    elif t == TP.OP_LN_BWD:
        dy, x = _mat(slots, p[0], M, N, ld[0]), _mat(slots, p[1], M, N, ld[1])
        w = _mat(slots, p[2], 1, N, N)[0]
        mean, rstd = _mat(slots, p[3], 1, M, M)[0], _mat(slots, p[4], 1, M, M)[0]
        xh = (x - mean[:, None]) * rstd[:, None]
        g = dy * w
        dx = rstd[:, None] * (g - g.mean(1, keepdims=True) - xh * (g * xh).mean(1, keepdims=True))
        if p[6]:
            tiles = -(-M // 16)
            part = _mat(slots, p[6], tiles, 2 * N, 2 * N)
            for tl in range(tiles):
                r = slice(tl * 16, min(M, tl * 16 + 16))
                part[tl, :N] = (dy[r] * xh[r]).sum(0)
                part[tl, N:] = dy[r].sum(0)
        _mat(slots, p[5], M, N, ld[2])[...] = dx.astype(np.float32)
    elif t in (TP.OP_ATTN_FWD, TP.OP_ATTN_BWD):
        Nq, Nk, d, H = M, N, K, int(op["i0"])
        nb = int(op["ntiles"]) // H
        for tile in range(nb * H):
            b, h = divmod(tile, H)
            q = _mat(slots, p[0] + 4 * (b * ld[4] + h * d), Nq, d, ld[0])
            k = _mat(slots, p[1] + 4 * (b * ld[5] + h * d), Nk, d, ld[1])
            v = _mat(slots, p[2] + 4 * (b * ld[6] + h * d), Nk, d, ld[2])
            if t == TP.OP_ATTN_FWD:
                s = (q @ k.T) * f0
                e = np.exp(s - s.max(1, keepdims=True))
                P = (e / e.sum(1, keepdims=True)).astype(np.float32)
                _mat(slots, p[4] + 4 * tile * Nq * Nk, Nq, Nk, Nk)[...] = P
                _mat(slots, p[3] + 4 * (b * ld[7] + h * d), Nq, d, ld[3])[...] = P @ v
            else:
                P = _mat(slots, p[3] + 4 * tile * Nq * Nk, Nq, Nk, Nk)
                do = _mat(slots, p[4] + 4 * (b * ld[7] + h * d), Nq, d, ld[3])
                dv = P.T @ do
                dp = do @ v.T
                ds = f0 * P * (dp - (dp * P).sum(1, keepdims=True))
                _mat(slots, p[5] + 4 * (b * Nq * ld[8] + h * d), Nq, d, ld[8])[...] = ds @ k
                _mat(slots, p[6] + 4 * (b * Nk * ld[9] + h * d), Nk, d, ld[9])[...] = ds.T @ q
                _mat(slots, p[7] + 4 * (b * Nk * ld[10] + h * d), Nk, d, ld[10])[...] = dv
    elif t == TP.OP_COPY2D:
        v = np.zeros((M, N), np.float32)
        if p[0]:
            per = int(op["i1"])
            v = v + _wrap_rows(_mat(slots, p[0], per if per > 0 else M, N, ld[0]), M, per)
        if p[1]:
            per = int(op["i2"])
            v = v + _wrap_rows(_mat(slots, p[1], per if per > 0 else M, N, ld[1]), M, per)
        out = _mat(slots, p[2], M, N, ld[2])
        out[...] = out + v if fl & TP.F_ACCUM else v
    else:
        raise ValueError(f"unknown op type {t}")


def run(packed, slots, reverse=False):
    """Execute a packed program (Prog.pack()) on host memory; `slots`: eight base addresses, slots[0] = 0.  reverse: the ops of every phase in
    reverse order — the device runs a phase's ops concurrently, so the result must not depend on their order."""
    ops, phase_ops, phase_tiles, _ = packed
    assert slots[0] == 0
    for ph in range(len(phase_tiles)):
        idx = range(int(phase_ops[ph]), int(phase_ops[ph + 1]))
        for i in (reversed(idx) if reverse else idx):
            run_op(ops[i], slots)


def check_phase_hazards(packed):
    """Within one phase no op may write what another op of the phase reads or writes (operand address ranges; accumulating ops read their target).
    -> number of (phase, op, op) conflicts.  Address-range based, so it is independent of the builder's own root / part bookkeeping."""
    ops, phase_ops, phase_tiles, _ = packed

    def rng(addr, rows, cols, ld):
        return (int(addr), int(addr) + 4 * ((rows - 1) * ld + cols), rows, cols, ld) if addr and rows > 0 and cols > 0 else None

    def rw(op):
        t, fl = int(op["type"]), int(op["flags"])
        M, N, K = int(op["M"]), int(op["N"]), int(op["K"])
        ld, p = [int(v) for v in op["ld"]], [int(v) for v in op["p"]]
        R, W = [], []
        if t == TP.OP_GEMM:
            ta, tb, s = fl & TP.F_TRANS_A, fl & TP.F_TRANS_B, max(int(op["i0"]), 1)
            R += [rng(p[0], K if ta else M, M if ta else K, ld[0]), rng(p[1], N if tb else K, K if tb else N, ld[1]), rng(p[3], 1, N, N),
                  rng(p[6], M, N, ld[3]), rng(p[7], M, N, ld[4])]
            W += [rng(p[2], M * s, (N - 1) * max(ld[9], 1) + 1, ld[2]), rng(p[8], M, N, ld[5]), rng(p[9], 1, M * s, M * s)]
        elif t == TP.OP_REDUCE:
            R += [rng(p[0] + 4 * s * ld[2], M, N, ld[0]) for s in range(K)] + [rng(p[2], 1, N, N), rng(p[3], M, N, ld[3]), rng(p[4], M, N, ld[4])]
            W += [rng(p[1], M, (N - 1) * max(ld[5], 1) + 1, ld[1])]
        elif t == TP.OP_LN_FWD:
            R += [rng(p[0], M, N, ld[0])]
            W += [rng(p[3], M, N, ld[1]), rng(p[4], 1, M, M), rng(p[5], 1, M, M)]
        elif t == TP.OP_LN_BWD:
            R += [rng(p[0], M, N, ld[0]), rng(p[1], M, N, ld[1]), rng(p[3], 1, M, M), rng(p[4], 1, M, M)]
            W += [rng(p[5], M, N, ld[2]), rng(p[6], -(-M // 16), 2 * N, 2 * N)]
        elif t in (TP.OP_ATTN_FWD, TP.OP_ATTN_BWD):
            H = int(op["i0"]); nb = int(op["ntiles"]) // H
            R += [rng(p[0], nb * M, H * K, ld[0]), rng(p[1], nb * N, H * K, ld[1]), rng(p[2], nb * N, H * K, ld[2])]
            if t == TP.OP_ATTN_FWD:
                W += [rng(p[3], nb * M, H * K, ld[3]), rng(p[4], nb * H * M, N, N)]
            else:
                R += [rng(p[3], nb * H * M, N, N), rng(p[4], nb * M, H * K, ld[3])]
                W += [rng(p[5], nb * M, H * K, ld[8]), rng(p[6], nb * N, H * K, ld[9]), rng(p[7], nb * N, H * K, ld[10])]
        elif t == TP.OP_COPY2D:
            W += [rng(p[2], M, N, ld[2])]
            R += [rng(p[0], int(op["i1"]) or M, N, ld[0]), rng(p[1], int(op["i2"]) or M, N, ld[1])]
        return [r for r in R if r], [w for w in W if w]

    def overlap(a, b):
        if not (a[0] < b[1] and b[0] < a[1]):
            return False
        if (a[4] == a[3] and b[4] == b[3]) or a[2] * b[2] > 65536:           # both dense (or too many rows to enumerate: assume they meet)
            return True
        sa = a[0] + 4 * a[4] * np.arange(a[2])[:, None]                       # row views (e.g. one token of every prompt): row by row
        sb = b[0] + 4 * b[4] * np.arange(b[2])[None, :]
        return bool(((sa < sb + 4 * b[3]) & (sb < sa + 4 * a[3])).any())
    bad = 0
    for ph in range(len(phase_tiles)):
        sets = [rw(ops[i]) for i in range(int(phase_ops[ph]), int(phase_ops[ph + 1]))]
        for i in range(len(sets)):
            for j in range(len(sets)):
                if i == j:
                    continue
                for w in sets[i][1]:
                    if any(overlap(w, r) for r in sets[j][0]) or (i < j and any(overlap(w, w2) for w2 in sets[j][1])):
                        bad += 1
    return bad
