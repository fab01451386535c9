"""The trainable fp32 mask tail lowered to ONE launch per direction (kernel: csrc/tail_program.hip).

What is lowered: text_hidden_fcs (model/MedPLIB.py:152-164) -> token assembly and `src = image_embeddings + dense` (mask_decoder.py:118-131)
-> TwoWayTransformer (transformer.py:62-106: two TwoWayAttentionBlocks :151-182 + the final token-to-image attention; Attention :185-244)
-> output_hypernetworks_mlps[0] and iou_prediction_head (mask_decoder.py:141-153, multimask_output=False), and the complete backward of that
chain (every weight / bias / LayerNorm / token gradient, the gradient of the <SEG> hidden rows).  The upsampler, `postprocess_masks` and the
losses stay kernels of their own (upsampler_fused.hip, mask_ops.hip).

How: the module graph is written down ONCE per (prompt count, parameter addresses) as a list of tile operations on named buffers
(`Prog`); every op declares what it reads and writes, the phase of an op is the earliest one that respects read-after-write,
write-after-write and write-after-read on those buffers, and the persistent kernel runs phase after phase with a grid barrier in between.
The backward is generated from a tape the forward lowering leaves behind (one closure per forward op, run in reverse), so the two
programs cannot drift apart.  Sums have a fixed order everywhere (split-K partials and column sums are combined by REDUCE ops).

The op table is plain data (numpy structured array, 256 bytes per op); `oracle/tail_program_emu.py` interprets the same table on the CPU
for the tests.  There is no CPU execution path here: `TailProgram.run_*` launch the HIP kernel and nothing else."""
import ctypes

import numpy as np
import torch

OP_GEMM, OP_REDUCE, OP_LN_FWD, OP_LN_BWD, OP_ATTN_FWD, OP_ATTN_BWD, OP_COPY2D = 1, 2, 3, 4, 5, 6, 7
F_TRANS_A, F_TRANS_B, F_RELU, F_ACCUM, F_CS_ACCUM, F_A_BF16 = 1, 2, 4, 8, 16, 32
OP_NAMES = {1: "gemm", 2: "reduce", 3: "ln_fwd", 4: "ln_bwd", 5: "attn_fwd", 6: "attn_bwd", 7: "copy2d"}
OP_DTYPE = np.dtype([("type", "<i4"), ("flags", "<i4"), ("ntiles", "<i4"), ("tile_begin", "<i4"), ("M", "<i4"), ("N", "<i4"), ("K", "<i4"),
                     ("i0", "<i4"), ("i1", "<i4"), ("i2", "<i4"), ("i3", "<i4"), ("pad0", "<i4"), ("f0", "<f4"), ("f1", "<f4"), ("f2", "<f4"),
                     ("f3", "<f4"), ("ld", "<i8", (12,)), ("p", "<u8", (12,))])
assert OP_DTYPE.itemsize == 256
POOL = 16384 - 32                # floats of LDS per workgroup the op tiles may use (tail_program.hip)
# slots of the operand address space (slot 0 = absolute addresses: parameters, constants)
S_ABS, S_FWD, S_BWD, S_IN0, S_GRAD, S_IN1, S_IN2, S_IN3 = range(8)


def _cdiv(a, b):
    return -(-a // b)


class Ref:
    """A 2-D fp32 operand: (slot, byte offset), rows x cols with a row stride in floats; `root` / `part` identify the buffer for hazards
    (two Refs conflict when they share the root and either covers the whole buffer or both name the same part)."""
    __slots__ = ("slot", "off", "rows", "cols", "ld", "root", "part", "relu")

    def __init__(self, slot, off, rows, cols, ld=None, root=None, part=None):
        self.slot, self.off, self.rows, self.cols = slot, int(off), int(rows), int(cols)
        self.ld = int(cols if ld is None else ld)
        self.root = root if root is not None else (slot, self.off)
        self.part = part
        self.relu = False

    @property
    def addr(self):
        assert 0 <= self.off < (1 << 56)
        return (self.slot << 56) | self.off

    def view(self, r0=0, rows=None, c0=0, cols=None, row_step=1, part=None):
        v = Ref(self.slot, self.off + 4 * (r0 * self.ld + c0), self.rows if rows is None else rows, self.cols if cols is None else cols,
                self.ld * row_step, self.root, self.part if part is None else part)
        return v

    @staticmethod
    def of(t, rows=None, cols=None):
        """An absolute operand for a tensor that outlives the program (parameter, constant)."""
        assert t.dtype == torch.float32 and t.is_contiguous()
        if rows is None:
            rows, cols = (t.numel() // t.shape[-1], t.shape[-1]) if t.dim() >= 2 else (1, t.numel())
        return Ref(S_ABS, t.data_ptr(), rows, cols)


class Arena:
    def __init__(self, slot):
        self.slot, self.size, self.names = slot, 0, {}

    def alloc(self, rows, cols, name):
        off = self.size
        self.size += _cdiv(rows * cols * 4, 256) * 256
        self.names[name] = off
        return Ref(self.slot, off, rows, cols, cols, root=(self.slot, off))


class Prog:
    """Ops in program order with sequential semantics; phases by buffer hazards."""

    def __init__(self):
        self.ops = []                    # (phase, index, fields dict)
        self._w, self._r = {}, {}        # root -> {part: last phase}

    @staticmethod
    def _hits(table, ref):
        d = table.get(ref.root)
        if not d:
            return -1
        if ref.part is None:
            return max(d.values())
        return max(d.get(ref.part, -1), d.get(None, -1))

    def _emit(self, fields, reads, writes):
        ph = 0
        for r in reads:
            ph = max(ph, self._hits(self._w, r) + 1)
        for w in writes:
            ph = max(ph, self._hits(self._w, w) + 1, self._hits(self._r, w) + 1)
        for r in reads:
            d = self._r.setdefault(r.root, {})
            d[r.part] = max(d.get(r.part, -1), ph)
        for w in writes:
            self._w.setdefault(w.root, {})[w.part] = ph
        self.ops.append((ph, len(self.ops), fields))
        return ph

    @staticmethod
    def _a(ref):
        return 0 if ref is None else ref.addr

    # ------------------------------------------------------------------ op constructors
    def gemm(self, A, B, C, ta=False, tb=False, a2=None, a2_rows=0, b2=None, b2_rows=0, bias=None, res=None, relu=False, mask=None,
             accum=False, c2=None, colsum=None, cs_accum=False, alpha=1.0, splits=1, a_bf16=False, c_cs=0, note=""):
        M, K = (A.cols, A.rows) if ta else (A.rows, A.cols)
        Kb, N = (B.cols, B.rows) if tb else (B.rows, B.cols)
        assert K == Kb, f"gemm {note}: inner dims {K} vs {Kb}"
        if splits > 1:
            assert C.rows == splits * M and C.cols == N and C.ld == N, f"gemm {note}: partial buffer shape"
            assert not (bias or res or relu or mask or accum or c2), "split GEMM: the epilogue belongs to the REDUCE"
            per = _cdiv(_cdiv(K, splits), 64) * 64
            assert per * (splits - 1) < K, f"gemm {note}: an empty K split"
        else:
            assert (C.rows, C.cols) == (M, N), f"gemm {note}: C is {C.rows}x{C.cols}, product is {M}x{N}"
        for x, sh in ((bias, (1, N)), (res, (M, N)), (mask, (M, N)), (c2, (M, N))):
            assert x is None or (x.rows, x.cols) == sh, f"gemm {note}: epilogue operand shape {(x.rows, x.cols)} vs {sh}"
        assert colsum is None or (ta and colsum.cols == M)
        flags = ((F_TRANS_A if ta else 0) | (F_TRANS_B if tb else 0) | (F_RELU if relu else 0) | (F_ACCUM if accum else 0) | (F_CS_ACCUM if cs_accum else 0)
                 | (F_A_BF16 if a_bf16 else 0))
        assert not (a_bf16 and a2 is not None) and not (c_cs and (splits > 1 or c2 is not None))
        ld = [A.ld, B.ld, C.ld, res.ld if res else 0, mask.ld if mask else 0, c2.ld if c2 else 0, a2.ld if a2 else 0, b2.ld if b2 else 0,
              M * N if splits > 1 else 0, c_cs, 0, 0]
        p = [self._a(A), self._a(B), self._a(C), self._a(bias), self._a(a2), self._a(b2), self._a(res), self._a(mask), self._a(c2), self._a(colsum), 0, 0]
        f = dict(type=OP_GEMM, flags=flags, ntiles=_cdiv(M, 64) * _cdiv(N, 64) * splits, M=M, N=N, K=K, i0=splits, i1=a2_rows, i2=b2_rows,
                 f0=alpha, ld=ld, p=p, note=note)
        reads = [x for x in (A, B, a2, b2, bias, res, mask) if x is not None] + ([C] if accum else []) + ([c2] if c2 else []) + \
                ([colsum] if (colsum is not None and cs_accum) else [])
        writes = [C] + ([c2] if c2 else []) + ([colsum] if colsum is not None else [])
        return self._emit(f, reads, writes)

    def reduce(self, src, out, S, split_stride, M=None, N=None, bias=None, res=None, relu=False, mask=None, accum=False, alpha=1.0, ld_in=None, out_cs=0,
               note=""):
        M = out.rows if M is None else M
        N = out.cols if N is None else N
        ld = [src.ld if ld_in is None else ld_in, out.ld, split_stride, res.ld if res else 0, mask.ld if mask else 0, out_cs] + [0] * 6
        p = [self._a(src), self._a(out), self._a(bias), self._a(res), self._a(mask)] + [0] * 7
        flags = (F_RELU if relu else 0) | (F_ACCUM if accum else 0)
        f = dict(type=OP_REDUCE, flags=flags, ntiles=_cdiv(M * N, 1024), M=M, N=N, K=S, f0=alpha, ld=ld, p=p, note=note)
        reads = [x for x in (src, bias, res, mask) if x is not None] + ([out] if accum else [])
        return self._emit(f, reads, [out])

    def ln_fwd(self, x, w, b, y, mean, rstd, eps, note=""):
        assert x.cols % 64 == 0 and x.cols <= 512
        f = dict(type=OP_LN_FWD, ntiles=_cdiv(x.rows, 16), M=x.rows, N=x.cols, f0=eps, ld=[x.ld, y.ld] + [0] * 10,
                 p=[x.addr, w.addr, b.addr, y.addr, mean.addr, rstd.addr] + [0] * 6, note=note)
        return self._emit(f, [x, w, b], [y, mean, rstd])

    def ln_bwd(self, dy, x, w, mean, rstd, dx, part, note=""):
        assert x.cols % 64 == 0 and x.cols <= 512
        f = dict(type=OP_LN_BWD, ntiles=_cdiv(x.rows, 16), M=x.rows, N=x.cols, ld=[dy.ld, x.ld, dx.ld] + [0] * 9,
                 p=[dy.addr, x.addr, w.addr, mean.addr, rstd.addr, dx.addr, self._a(part)] + [0] * 5, note=note)
        return self._emit(f, [dy, x, w, mean, rstd], [dx] + ([part] if part is not None else []))

    @staticmethod
    def _attn_check(Nq, Nk, d, bwd):
        need = (2 * (Nq + Nk) * (d + 1) + 2 * Nq * Nk) if bwd else ((Nq + 2 * Nk) * (d + 1) + Nq * Nk)
        assert need <= POOL, f"attention core {Nq}x{Nk}x{d} does not fit the LDS pool"

    def attn_fwd(self, q, k, v, o, P, n, H, Nq, Nk, scale, note=""):
        d = q.cols // H
        self._attn_check(Nq, Nk, d, False)
        ld = [q.ld, k.ld, v.ld, o.ld, Nq * q.ld, Nk * k.ld, Nk * v.ld, Nq * o.ld] + [0] * 4
        f = dict(type=OP_ATTN_FWD, ntiles=n * H, M=Nq, N=Nk, K=d, i0=H, f0=scale, ld=ld, p=[q.addr, k.addr, v.addr, o.addr, P.addr] + [0] * 7, note=note)
        return self._emit(f, [q, k, v], [o, P])

    def attn_bwd(self, q, k, v, P, do, dq, dk, dv, n, H, Nq, Nk, scale, note=""):
        d = q.cols // H
        self._attn_check(Nq, Nk, d, True)
        ld = [q.ld, k.ld, v.ld, do.ld, Nq * q.ld, Nk * k.ld, Nk * v.ld, Nq * do.ld, dq.ld, dk.ld, dv.ld, 0]
        f = dict(type=OP_ATTN_BWD, ntiles=n * H, M=Nq, N=Nk, K=d, i0=H, f0=scale, ld=ld,
                 p=[q.addr, k.addr, v.addr, P.addr, do.addr, dq.addr, dk.addr, dv.addr] + [0] * 4, note=note)
        return self._emit(f, [q, k, v, P, do], [dq, dk, dv])

    def copy2d(self, out, a=None, b=None, a_rows=0, b_rows=0, accum=False, note=""):
        f = dict(type=OP_COPY2D, flags=F_ACCUM if accum else 0, ntiles=_cdiv(out.rows * out.cols, 1024), M=out.rows, N=out.cols, i1=a_rows, i2=b_rows,
                 ld=[a.ld if a else 0, b.ld if b else 0, out.ld] + [0] * 9, p=[self._a(a), self._a(b), out.addr] + [0] * 9, note=note)
        return self._emit(f, [x for x in (a, b) if x is not None] + ([out] if accum else []), [out])

    # ------------------------------------------------------------------ packing
    def pack(self):
        """-> (ops structured array sorted by phase, phase_ops [P + 1], phase_tiles [P], notes)."""
        order = sorted(self.ops, key=lambda t: (t[0], t[1]))
        n_ph = (order[-1][0] + 1) if order else 0
        arr = np.zeros(len(order), dtype=OP_DTYPE)
        phase_ops = np.zeros(n_ph + 1, dtype=np.int32)
        phase_tiles = np.zeros(n_ph, dtype=np.int32)
        notes = []
        for i, (ph, _, f) in enumerate(order):
            rec = arr[i]
            for k, v in f.items():
                if k == "note":
                    continue
                rec[k] = v
            rec["tile_begin"] = phase_tiles[ph]
            phase_tiles[ph] += f["ntiles"]
            phase_ops[ph + 1] = i + 1
            notes.append((ph, OP_NAMES[f["type"]], f.get("note", "")))
        for ph in range(n_ph):                               # phases are contiguous by construction; an empty one would repeat the bound
            phase_ops[ph + 1] = max(phase_ops[ph + 1], phase_ops[ph])
        return arr, phase_ops, phase_tiles, notes


def _splits(tiles, K, target=256):
    """Split-K count for a GEMM with `tiles` output tiles and a reduction of K: enough units for the grid, >= 128 deep, whole 64-slabs."""
    if K < 512:
        return 1
    for s in (16, 8, 4, 2):
        if tiles * s <= target and K % (s * 64) == 0 and K // s >= 128:
            return s
    return 1


class _Lowering:
    """Forward lowering + tape.  `fw` / `bw` are the two programs, `wf` the forward workspace (activations kept for the backward), `wb` the
    backward scratch.  Gradients of activations live in `wb`, keyed by the activation's root."""

    def __init__(self, n, grad_of):
        self.n = n
        self.fw, self.bw = Prog(), Prog()
        self.wf, self.wb = Arena(S_FWD), Arena(S_BWD)
        self.tape = []
        self.g = {}                    # activation root -> gradient Ref (complete once every consumer's closure has run)
        self.grad_of = grad_of         # parameter tensor -> gradient Ref (slot S_GRAD) or None when the parameter is frozen
        self.nograd = set()            # activations nothing trainable sits in front of (src0 = frozen image embedding + frozen dense prompt)
        self._uid = 0

    def _name(self, s):
        self._uid += 1
        return f"{s}#{self._uid}"

    # ---- gradient buffers -------------------------------------------------------------------------------------------
    def grad(self, x):
        """(gradient Ref of activation x shaped like x, already_written)."""
        key = (x.root, x.off, x.rows, x.cols, x.ld)
        if key in self.g:
            return self.g[key], True
        r = self.wb.alloc(x.rows, x.cols, self._name("g"))
        self.g[key] = r
        return r, False

    def grad_alias(self, x, ref):
        key = (x.root, x.off, x.rows, x.cols, x.ld)
        assert key not in self.g
        self.g[key] = ref

    def has_grad(self, x):
        return (x.root, x.off, x.rows, x.cols, x.ld) in self.g

    def set_grad(self, x, ref):
        self.g[(x.root, x.off, x.rows, x.cols, x.ld)] = ref

    # ---- layers -----------------------------------------------------------------------------------------------------
    def linear(self, x, w_t, b_t, name, pe=None, pe_rows=0, pe_grad=None, relu=False, res=None, x_needs_grad=True):
        """y = act((x [+ pe]) W^T + b) [+ res].  pe_grad: the activation whose gradient also receives d(x + pe) (the token embedding used as
        query_pe), None for constants.  Backward: dX (+ into pe's gradient), dW, db; the residual's gradient is the output's."""
        fw, M, K, N = self.fw, x.rows, x.cols, w_t.shape[0]
        W, Bv = Ref.of(w_t), Ref.of(b_t, 1, N)
        y = self.wf.alloc(M, N, name)
        s = _splits(_cdiv(M, 64) * _cdiv(N, 64), K)
        if s > 1:
            part = self.wf.alloc(s * M, N, name + ".part")
            fw.gemm(x, W, part, tb=True, a2=pe, a2_rows=pe_rows, splits=s, note=name)
            fw.reduce(part, y, s, M * N, bias=Bv, res=res, relu=relu, ld_in=N, note=name + ".sum")
        else:
            fw.gemm(x, W, y, tb=True, a2=pe, a2_rows=pe_rows, bias=Bv, res=res, relu=relu, note=name)
        y.relu = relu

        def back():
            bw = self.bw
            dy, ok = self.grad(y)
            assert ok, f"{name}: no gradient reached the output"
            if res is not None:                                   # y = res + ...: the residual's gradient is dy
                if self.has_grad(res):
                    gr, _ = self.grad(res)
                    bw.copy2d(gr, a=dy, accum=True, note=name + ".dres")
                else:
                    self.grad_alias(res, dy)
            gw, gb = self.grad_of(w_t), self.grad_of(b_t)
            if gw is not None:                                    # dW [N, K] = dy^T (x + pe), db = column sums of dy: one pass over dy
                sw = _splits(_cdiv(N, 64) * _cdiv(K, 64), M)
                if sw > 1:
                    pw = self.wb.alloc(sw * N, K, self._name(name + ".dWpart"))
                    pb = self.wb.alloc(sw, N, self._name(name + ".dbpart")) if gb is not None else None
                    bw.gemm(dy, x, pw, ta=True, b2=pe, b2_rows=pe_rows, colsum=pb.view(0, 1, 0, N) if pb else None, splits=sw, note=name + ".dW")
                    bw.reduce(pw, gw, sw, N * K, accum=True, ld_in=K, note=name + ".dW.sum")
                    if pb is not None:
                        bw.reduce(pb, gb, sw, N, M=1, N=N, accum=True, ld_in=N, note=name + ".db.sum")
                else:
                    bw.gemm(dy, x, gw, ta=True, b2=pe, b2_rows=pe_rows, accum=True, colsum=gb, cs_accum=True, note=name + ".dW")
            if x_needs_grad and (x.root, x.off, x.rows, x.cols, x.ld) not in self.nograd:   # dX [M, K] = dy W, masked by the ReLU that produced x
                gx, written = self.grad(x)
                mask = x if x.relu else None
                assert not (mask is not None and written), f"{name}: a ReLU output with two consumers"
                gpe = None
                if pe_grad is not None:
                    gpe, okp = self.grad(pe_grad)
                    assert okp, "the token-gradient buffer is zero-filled before its first consumer"
                sx = _splits(_cdiv(M, 64) * _cdiv(K, 64), N)
                if sx > 1:
                    px = self.wb.alloc(sx * M, K, self._name(name + ".dXpart"))
                    bw.gemm(dy, W, px, splits=sx, note=name + ".dX")
                    if gpe is not None:
                        bw.reduce(px, gpe, sx, M * K, accum=True, ld_in=K, note=name + ".dpe.sum")
                    bw.reduce(px, gx, sx, M * K, mask=mask, accum=written, ld_in=K, note=name + ".dX.sum")
                else:
                    bw.gemm(dy, W, gx, mask=mask, accum=written, c2=gpe, note=name + ".dX")
        self.tape.append(back)
        return y

    def layernorm(self, x, ln, name):
        fw = self.fw
        Wv, Bv = Ref.of(ln.weight, 1, x.cols), Ref.of(ln.bias, 1, x.cols)
        y = self.wf.alloc(x.rows, x.cols, name)
        mean, rstd = self.wf.alloc(1, x.rows, name + ".mean"), self.wf.alloc(1, x.rows, name + ".rstd")
        fw.ln_fwd(x, Wv, Bv, y, mean, rstd, float(ln.eps), note=name)

        def back():
            bw = self.bw
            dy, ok = self.grad(y)
            assert ok, f"{name}: no gradient reached the output"
            dx, written = self.grad(x)
            assert not written, f"{name}: the pre-norm sum has one consumer"
            gw, gb = self.grad_of(ln.weight), self.grad_of(ln.bias)
            tiles = _cdiv(x.rows, 16)
            part = self.wb.alloc(tiles, 2 * x.cols, self._name(name + ".wb")) if gw is not None else None
            bw.ln_bwd(dy, x, Wv, mean, rstd, dx, part, note=name + ".bwd")
            if part is not None:
                bw.reduce(part.view(0, tiles, 0, x.cols), gw, tiles, 2 * x.cols, M=1, N=x.cols, accum=True, ld_in=2 * x.cols, note=name + ".dw")
                bw.reduce(part.view(0, tiles, x.cols, x.cols), gb, tiles, 2 * x.cols, M=1, N=x.cols, accum=True, ld_in=2 * x.cols, note=name + ".db")
        self.tape.append(back)
        return y

    def attention(self, att, name, xq, xk, xv, Nq, Nk, q_pe=None, q_pe_rows=0, q_pe_grad=None, k_pe=None, k_pe_rows=0, k_pe_grad=None, res=None):
        """Attention.forward (transformer.py:224-244): out_proj(softmax(q k^T / sqrt(d)) v) [+ res] with q = (xq + q_pe) Wq, k = (xk + k_pe) Wk, v = xv Wv."""
        n, H = self.n, att.num_heads
        q = self.linear(xq, att.q_proj.weight, att.q_proj.bias, name + ".q", pe=q_pe, pe_rows=q_pe_rows, pe_grad=q_pe_grad)
        k = self.linear(xk, att.k_proj.weight, att.k_proj.bias, name + ".k", pe=k_pe, pe_rows=k_pe_rows, pe_grad=k_pe_grad)
        v = self.linear(xv, att.v_proj.weight, att.v_proj.bias, name + ".v")
        inner = q.cols
        scale = 1.0 / float(np.sqrt(inner // H))
        o = self.wf.alloc(n * Nq, inner, name + ".o")
        P = self.wf.alloc(n * H * Nq, Nk, name + ".P")
        self.fw.attn_fwd(q, k, v, o, P, n, H, Nq, Nk, scale, note=name + ".core")

        def back():
            do, ok = self.grad(o)
            assert ok
            dq, w1 = self.grad(q); dk, w2 = self.grad(k); dv, w3 = self.grad(v)
            assert not (w1 or w2 or w3)
            self.bw.attn_bwd(q, k, v, P, do, dq, dk, dv, n, H, Nq, Nk, scale, note=name + ".core.bwd")
        self.tape.append(back)
        return self.linear(o, att.out_proj.weight, att.out_proj.bias, name + ".out", res=res)

    def mlp3(self, x, mlp, name):
        k = len(mlp.layers)
        for i, l in enumerate(mlp.layers):
            x = self.linear(x, l.weight, l.bias, f"{name}.{i}", relu=i < k - 1)
        return x


class TailProgram:
    """The two programs for `n` prompts of one MaskDecoder (+ optionally text_hidden_fcs in front of it), bound to the CURRENT addresses of
    the parameters, their gradients (flat buffer offsets) and the two constants.  Rebuilt by `get_program` when any of those move."""

    def __init__(self, dec, n, dense_pe, no_mask_embed, fcs=None, grad_offsets=None, hidden_grad=True, text_grad=True, fused_upsampler=False):
        """dec: model.sam.MaskDecoder; fcs: (fc1 Linear, fc2 Linear) or None (then the text embedding [n, C] is the input);
        grad_offsets: {id(param): byte offset into the gradient buffer the backward is given} for the parameters that train.
        fused_upsampler: the backward's inputs are the outputs of mp_mask_upsample_fused_bwd_bf16 in ONE buffer (ops.upsample_bwd_layout) and
        the bf16 tokens the upsampler consumed — the program then also finishes that kernel's work: d src = the sum of its two halves,
        d hyper0 and the bias / LayerNorm2d gradients from its per-task rows, and the two ConvTranspose2d weight gradients (dW1 = dy1^T src,
        dW2 = dy2^T a1) written straight into the [Cin, Cout, 2, 2] gradient tensors (mask_decoder.py:53-59)."""
        self.n, self.C, self.Tk = n, dec.dim, dec.grid * dec.grid
        self.dec = dec
        self.device = dec.iou_token.weight.device
        n, C, Tk = self.n, self.C, self.Tk
        grad_offsets = grad_offsets or {}
        self.grad_offsets = dict(grad_offsets)

        def grad_of(t):
            off = grad_offsets.get(id(t))
            if off is None:
                return None
            rows, cols = (t.numel() // t.shape[-1], t.shape[-1]) if t.dim() >= 2 else (1, t.numel())
            return Ref(S_GRAD, off, rows, cols)
        L = self.L = _Lowering(n, grad_of)
        fw, tr = L.fw, dec.transformer
        kpe = Ref.of(dense_pe)                                                     # [Tk, C], a model constant (prompt_encoder.py:62-71)
        assert (kpe.rows, kpe.cols) == (Tk, C)
        # ---- inputs
        if fcs is not None:
            Dh = fcs[0].weight.shape[1]
            hid_in = Ref(S_IN0, 0, n, Dh)
            hid = L.wf.alloc(n, Dh, "hidden_rows")                                  # kept: dW of fc1 needs it
            fw.copy2d(hid, a=hid_in, note="hidden_rows")
            f1 = L.linear(hid, fcs[0].weight, fcs[0].bias, "fc1", relu=True, x_needs_grad=hidden_grad)
            text = L.linear(f1, fcs[1].weight, fcs[1].bias, "fc2")
            self.in_dim = Dh
        else:
            text_in = Ref(S_IN0, 0, n, C)
            text = L.wf.alloc(n, C, "text")
            fw.copy2d(text, a=text_in, note="text")
            self.in_dim = C
        img_in = Ref(S_IN1, 0, n * Tk, C)
        # tokens = cat([iou_token; mask_tokens] per prompt, text) (mask_decoder.py:123-125); src = image_embeddings + dense (:128-129)
        tok = L.wf.alloc(n * 6, C, "tokens")
        fw.copy2d(tok.view(0, n, 0, C, row_step=6, part="r0"), a=Ref.of(dec.iou_token.weight), a_rows=1, note="tok.iou")
        for j in range(4):
            fw.copy2d(tok.view(1 + j, n, 0, C, row_step=6, part=f"r{1 + j}"), a=Ref.of(dec.mask_tokens.weight).view(j, 1), a_rows=1, note=f"tok.mask{j}")
        fw.copy2d(tok.view(5, n, 0, C, row_step=6, part="r5"), a=text, note="tok.text")
        src = L.wf.alloc(n * Tk, C, "src0")
        fw.copy2d(src, a=img_in, b=Ref.of(no_mask_embed, 1, C), b_rows=1, note="src0")
        L.nograd.add(self._key(src))

        def tok_back():            # runs LAST in the backward: tokens were the initial queries AND the query_pe of every attention
            gt, ok = L.grad(tok)
            assert ok
            bw = L.bw
            g_iou, g_msk = grad_of(dec.iou_token.weight), grad_of(dec.mask_tokens.weight)
            if g_iou is not None:
                bw.reduce(gt.view(0, 1, 0, C), g_iou, n, 6 * C, M=1, N=C, accum=True, ld_in=C, note="d iou_token")
            if g_msk is not None:
                bw.reduce(gt.view(1, 4, 0, C), g_msk, n, 6 * C, M=4, N=C, accum=True, ld_in=C, note="d mask_tokens")
            L.set_grad(text, gt.view(5, n, 0, C, row_step=6))                      # d text = the sixth token's gradient rows
        L.tape.append(tok_back)

        # ---- TwoWayTransformer (transformer.py:62-106)
        queries, keys = tok, src
        for li, layer in enumerate(tr.layers):
            p = f"l{li}"
            if layer.skip_first_layer_pe:                                          # :151-182
                a1 = L.attention(layer.self_attn, p + ".self", queries, queries, queries, 6, 6)
            else:
                a1 = L.attention(layer.self_attn, p + ".self", queries, queries, queries, 6, 6, q_pe=tok, q_pe_grad=tok, k_pe=tok, k_pe_grad=tok,
                                 res=queries)
            q1 = L.layernorm(a1, layer.norm1, p + ".norm1")
            a2 = L.attention(layer.cross_attn_token_to_image, p + ".t2i", q1, keys, keys, 6, Tk, q_pe=tok, q_pe_grad=tok, k_pe=kpe, k_pe_rows=Tk,
                             res=q1)
            q2 = L.layernorm(a2, layer.norm2, p + ".norm2")
            h = L.linear(q2, layer.mlp.lin1.weight, layer.mlp.lin1.bias, p + ".mlp1", relu=True)
            a3 = L.linear(h, layer.mlp.lin2.weight, layer.mlp.lin2.bias, p + ".mlp2", res=q2)
            q3 = L.layernorm(a3, layer.norm3, p + ".norm3")
            a4 = L.attention(layer.cross_attn_image_to_token, p + ".i2t", keys, q3, q3, Tk, 6, q_pe=kpe, q_pe_rows=Tk, k_pe=tok, k_pe_grad=tok,
                             res=keys)
            keys = L.layernorm(a4, layer.norm4, p + ".norm4")
            queries = q3
        a5 = L.attention(tr.final_attn_token_to_image, "final", queries, keys, keys, 6, Tk, q_pe=tok, q_pe_grad=tok, k_pe=kpe, k_pe_rows=Tk, res=queries)
        hs = L.layernorm(a5, tr.norm_final_attn, "norm_final")
        # ---- heads (mask_decoder.py:141-153; multimask_output=False keeps mask token 0)
        iou_x = hs.view(0, n, 0, C, row_step=6, part="r0")
        msk_x = hs.view(1, n, 0, C, row_step=6, part="r1")
        hyper0 = L.mlp3(msk_x, dec.output_hypernetworks_mlps[0], "hyper0")
        iou4 = L.mlp3(iou_x, dec.iou_prediction_head, "iou")
        self.out = {"src": keys, "hyper0": hyper0, "iou4": iou4, "text": text}
        self.fwd_bytes = L.wf.size

        # ---- backward program: the seeds, then the tape in reverse
        bw = L.bw
        g_tok, _ = L.grad(tok)
        bw.copy2d(g_tok, note="zero d tokens")
        g_hs, _ = L.grad(hs)
        bw.copy2d(g_hs, note="zero d hs")
        # the head inputs are row views of hs: their "gradients" are the same views of d hs (written once each, disjoint rows)
        L.set_grad(iou_x, g_hs.view(0, n, 0, C, row_step=6, part="r0"))
        L.set_grad(msk_x, g_hs.view(1, n, 0, C, row_step=6, part="r1"))
        g_keys, _ = L.grad(keys)
        g_hy, _ = L.grad(hyper0)
        self.fused_upsampler = bool(fused_upsampler)
        if not fused_upsampler:
            d_src = Ref(S_IN1, 0, n * Tk, C)
            bw.copy2d(g_keys, a=d_src, b=Ref(S_IN1, 4 * n * Tk * C, n * Tk, C), note="d src = the two halves of the upsampler's dx")
            bw.copy2d(g_hy, a=Ref(S_IN2, 0, n, hyper0.cols), note="d hyper0")
        else:
            from . import ops
            assert C == 256 and hyper0.cols == 32 and Tk % 8 == 0
            offs, _ = ops.upsample_bwd_layout(n, Tk)
            rows1, rows4, tasks = n * Tk, 4 * n * Tk, n * Tk // 8

            def ub(k, rows, cols, ld=None, c0=0):
                return Ref(S_IN1, 4 * (offs[k] + c0), rows, cols, cols if ld is None else ld, root=(S_IN1, offs[k]))
            bw.copy2d(g_keys, a=ub(0, rows1, C), b=Ref(S_IN1, 4 * (offs[0] + rows1 * C), rows1, C), note="d src = dx2[0] + dx2[1]")
            # part [tasks, 256]: db1[64] | dln_w[64] | dln_b[64] | db2[32] | dhyper[32]; a prompt's tasks are consecutive rows
            bw.reduce(ub(4, n, 32, ld=(Tk // 8) * 256, c0=224), g_hy, Tk // 8, 256, M=n, N=32, note="d hyper0 = its prompt's task rows summed")
            up = dec.output_upscaling
            for t_, c0, w_ in ((up[0].bias, 0, 64), (up[1].weight, 64, 64), (up[1].bias, 128, 64), (up[3].bias, 192, 32)):
                gt_ = grad_of(t_)
                if gt_ is not None:
                    bw.reduce(ub(4, 1, w_, ld=256, c0=c0), gt_, tasks, 256, M=1, N=w_, accum=True, note="upsampler bias / LayerNorm2d gradient")
            src_bf = Ref(S_IN2, 0, rows1, C)                      # bf16 tokens: the ld is in ELEMENTS
            for wt, A_, a_bf, rowsK, co, ci, kB in ((up[0].weight, src_bf, True, rows1, 64, 256, 1), (up[3].weight, ub(2, rows4, 64), False, rows4, 32, 64, 3)):
                gw_ = grad_of(wt)
                if gw_ is None:
                    continue
                base = Ref(S_GRAD, gw_.off, ci, co * 4)           # the [Cin, Cout, 2, 2] gradient as [Cin, Cout * 4]
                sw = _splits(_cdiv(ci, 64) * _cdiv(co, 64), rowsK)
                for k in range(4):                                # one product per (kh, kw): column k of every (cin, cout) cell, column stride 4
                    B_ = ub(kB, rowsK, co, ld=4 * co, c0=k * co)
                    Ck = Ref(S_GRAD, gw_.off + 4 * k, ci, co, co * 4, root=base.root)
                    if sw > 1:
                        pw = L.wb.alloc(sw * ci, co, L._name("ups.dWpart"))
                        bw.gemm(A_, B_, pw, ta=True, splits=sw, a_bf16=a_bf, note=f"ups.dW{co}.{k}")
                        bw.reduce(pw, Ck, sw, ci * co, accum=True, ld_in=co, out_cs=4, note=f"ups.dW{co}.{k}.sum")
                    else:
                        bw.gemm(A_, B_, Ck, ta=True, accum=True, a_bf16=a_bf, c_cs=4, note=f"ups.dW{co}.{k}")
        g_iou, _ = L.grad(iou4)
        bw.copy2d(g_iou, note="zero d iou4")
        bw.copy2d(g_iou.view(0, n, 0, 1), a=Ref(S_IN3, 0, n, 1), accum=True, note="d iou[:, 0]")
        # head closures treat x = a view of hs: dX writes (not accumulates) into zero-filled rows -> `written` must read False for them
        for cb in reversed(L.tape):
            cb()
        self.d_in = L.g.get(self._key(hid)) if fcs is not None else L.g.get(self._key(text))
        if fcs is not None and not hidden_grad:
            self.d_in = None
        self.bwd_bytes = L.wb.size
        self.fwd_packed = fw.pack()
        self.bwd_packed = bw.pack()
        self._dev = None

    @staticmethod
    def _key(x):
        return (x.root, x.off, x.rows, x.cols, x.ld)

    # ------------------------------------------------------------------ device side
    def _upload(self):
        if self._dev is not None:
            return self._dev
        dev = self.device
        d = {}
        for tag, (arr, po, pt, _) in (("f", self.fwd_packed), ("b", self.bwd_packed)):
            d[tag] = (torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).to(dev), torch.from_numpy(po.copy()).to(dev),
                      torch.from_numpy(pt.copy()).to(dev), len(pt))
        d["sync"] = None                                            # the LAST launch's barrier words (every launch takes its own 128 bytes)
        d["wb"] = torch.empty(max(self.bwd_bytes, 256) // 4, dtype=torch.float32, device=dev)
        self._dev = d
        return d

    def _launch(self, tag, slots, grid, stamps=None):
        from . import ops
        d = self._upload()
        o, po, pt, nph = d[tag]
        arr = (ctypes.c_uint64 * 8)(*slots)
        self.check_previous()
        # the barrier words belong to the LAUNCH, not to the program: the forward of an eval pass on one stream and the backward of a training step on
        # another may walk the same cached program at once (the caching allocator hands out per-stream blocks; mp_tail_program_run zeroes them)
        sync = torch.empty(32, dtype=torch.int32, device=self.device)
        ops.lib().call("mp_tail_program_run", o.data_ptr(), po.data_ptr(), pt.data_ptr(), nph, ctypes.addressof(arr), sync.data_ptr(),
                       stamps.data_ptr() if stamps is not None else None, int(grid), ops._stream())
        d["sync"] = sync
        # the give-up word comes back without a host wait: one 4-byte copy into pinned memory behind the launch, read at a LATER launch
        # (at most one read-back in flight: the pinned word is shared)
        if self._pending is None:
            if self._flag_host is None:
                self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._flag_host.copy_(sync[1:2], non_blocking=True)
            self._pending = torch.cuda.Event()
            self._pending.record()

    _pending = None
    _flag_host = None

    def check_previous(self):
        """A barrier of an EARLIER launch of this program gave up (a workgroup of its grid never became resident): its results were void.  Raises —
        after switching the owning decoder to the op-by-op tail, so a caller that catches and repeats the step gets a working path."""
        ev = self._pending
        if ev is None or not ev.query():
            return
        self._pending = None
        if int(self._flag_host[0]) != 0:
            self._flag_host.zero_()
            self.dec.use_program = False
            raise RuntimeError("the mask tail's program launch gave up at a grid barrier (sync[1] != 0): the results of that step are void; "
                               "the decoder now runs the op-by-op tail (MP_TAIL_PROGRAM=0 selects it from the start)")

    def run_forward(self, x_in, image_tokens, grid=256, stamps=None):
        """x_in: hidden rows [n, Dh] (with fcs) or text embeddings [n, C]; image_tokens [n, Tk, C] fp32 -> (workspace, src [n, Tk, C],
        hyper0 [n, 32], iou4 [n, 4]) — the three results are views of the workspace, which the backward needs untouched."""
        assert x_in.dtype == torch.float32 and x_in.is_contiguous() and tuple(x_in.shape) == (self.n, self.in_dim), (x_in.shape, self.in_dim)
        assert image_tokens.dtype == torch.float32 and image_tokens.is_contiguous() and image_tokens.numel() == self.n * self.Tk * self.C
        ws = torch.empty(self.fwd_bytes // 4, dtype=torch.float32, device=self.device)
        self._launch("f", [0, ws.data_ptr(), 0, x_in.data_ptr(), 0, image_tokens.data_ptr(), 0, 0], grid, stamps)
        o = self.out

        def view(r):
            assert r.ld == r.cols
            return ws[r.off // 4: r.off // 4 + r.rows * r.cols].view(r.rows, r.cols)
        return ws, view(o["src"]).view(self.n, self.Tk, self.C), view(o["hyper0"]), view(o["iou4"])

    def run_backward(self, ws, d_src2, d_hyper0, d_iou, grad_base, grid=256, stamps=None):
        """d_src2 [2, n, Tk, C] (the fused upsampler's two dx halves; pass a zero second half for a single gradient), d_hyper0 [n, 32],
        d_iou [n]; grad_base: address of the gradient buffer the program's offsets refer to (accumulated into).  -> gradient of the input
        rows [n, in_dim] (a copy: the program's scratch is rewritten by the next backward) or None."""
        for t in (d_src2, d_hyper0, d_iou):
            assert t.dtype == torch.float32 and t.is_contiguous()
        assert d_src2.numel() == 2 * self.n * self.Tk * self.C and d_hyper0.numel() == self.n * self.out["hyper0"].cols and d_iou.numel() == self.n
        assert not self.fused_upsampler, "a fused-upsampler program takes run_backward_fused"
        d = self._upload()
        self._launch("b", [0, ws.data_ptr(), d["wb"].data_ptr(), 0, int(grad_base), d_src2.data_ptr(), d_hyper0.data_ptr(), d_iou.data_ptr()], grid, stamps)
        if self.d_in is None:
            return None
        r = self.d_in
        flat = d["wb"][r.off // 4: r.off // 4 + (r.rows - 1) * r.ld + r.cols]
        return flat.as_strided((r.rows, r.cols), (r.ld, 1)).clone()        # autograd may keep it: never a view of the program's scratch

    def run_backward_fused(self, ws, ups_buf, src_bf16, d_iou, grad_base, grid=256, stamps=None):
        """The fused-upsampler form: ups_buf = mask_upsample_fused_bwd's one-buffer outputs, src_bf16 = the bf16 tokens its forward read."""
        assert self.fused_upsampler and ups_buf.dtype == torch.float32 and src_bf16.dtype == torch.bfloat16 and src_bf16.is_contiguous()
        assert d_iou.dtype == torch.float32 and d_iou.is_contiguous() and d_iou.numel() == self.n
        d = self._upload()
        self._launch("b", [0, ws.data_ptr(), d["wb"].data_ptr(), 0, int(grad_base), ups_buf.data_ptr(), src_bf16.data_ptr(), d_iou.data_ptr()], grid, stamps)
        if self.d_in is None:
            return None
        r = self.d_in
        flat = d["wb"][r.off // 4: r.off // 4 + (r.rows - 1) * r.ld + r.cols]
        return flat.as_strided((r.rows, r.cols), (r.ld, 1)).clone()

    def check_sync(self):
        """True when no barrier of the last launches gave up (reads the device: tests only)."""
        sync = self._upload()["sync"]
        return sync is None or int(sync[1].item()) == 0


# ---------------------------------------------------------------------------------------------------------------------- autograd face
def _direct_still_valid(base, params, offs):
    """The 'direct' form writes through addresses taken at FORWARD time (base + the program's offsets = each trainable parameter's .grad).  Between
    forward and backward a caller may have dropped or re-homed the gradients (zero_grad(set_to_none=True), a re-flattening engine): then those
    addresses are not the parameters' gradients any more — possibly freed memory."""
    for p_ in params:
        if p_.requires_grad:
            g = p_.grad
            if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() != base + offs[id(p_)]:
                return False
    return True


def _detached_grads(prog_backward, params, offs, dev):
    """Fallback of the direct form when the .grad tensors moved: the same program (its offsets are relative to a base) run on a fresh zero buffer
    spanning the old layout; the parameter gradients go back to autograd as views of it."""
    span = max((offs[id(p_)] // 4 + p_.numel() for p_ in params if p_.requires_grad), default=0)
    gflat = torch.zeros(max(span, 64), dtype=torch.float32, device=dev)
    d_in = prog_backward(gflat.data_ptr())
    grads = [gflat[offs[id(p_)] // 4: offs[id(p_)] // 4 + p_.numel()].view(p_.shape) if p_.requires_grad else None for p_ in params]
    return d_in, grads


class TailProgramFn(torch.autograd.Function):
    """(x_in, image_tokens) -> (src [n, Tk, C], hyper0 [n, 32], iou [n]) through the forward program; the backward program writes every
    parameter gradient itself: straight into the parameters' `.grad` (the engine's flat buffer) when they all live in one buffer — autograd
    then sees None for them, so there is no per-parameter accumulation launch — or into one fresh zero buffer whose views are returned."""

    @staticmethod
    def forward(ctx, runner, x_in, image_tokens, *params):
        ctx.set_materialize_grads(False)
        prog, base, layout = runner.program(x_in.shape[0], params, x_in.requires_grad)
        ws, src, hyper0, iou4 = prog.run_forward(x_in.detach().contiguous(), image_tokens.detach().contiguous(), grid=runner.grid)
        ctx.prog, ctx.ws, ctx.runner, ctx.base, ctx.layout, ctx.n_params = prog, ws, runner, base, layout, len(params)
        ctx.params = params
        return src, hyper0, iou4[:, 0]

    @staticmethod
    def backward(ctx, d_src, d_hyper0, d_iou):
        prog, runner = ctx.prog, ctx.runner
        n, dev = prog.n, prog.device
        zeros = runner.zeros(n, dev)
        if d_src is None:
            d_src2 = zeros["src2"]
        elif d_src.dim() == 4:                                  # the fused upsampler's two halves, unsummed
            d_src2 = d_src.contiguous()
        else:                                                   # a single gradient tensor (the fp32 upscaling chain): first half, the second stays zero
            d_src2 = zeros["src2_one"]
            d_src2[0].copy_(d_src.reshape(d_src2[0].shape))
        d_hy = zeros["hy"] if d_hyper0 is None else d_hyper0.contiguous()
        d_io = zeros["iou"] if d_iou is None else d_iou.contiguous()
        if ctx.base is not None and not _direct_still_valid(ctx.base, ctx.params, runner.direct_offsets(ctx.params, ctx.base, prog)):
            d_in, grads = _detached_grads(lambda b: prog.run_backward(ctx.ws, d_src2, d_hy, d_io, b, grid=runner.grid), ctx.params, prog.grad_offsets, dev)
        elif ctx.base is not None:                              # direct: accumulate into the flat gradient buffer
            d_in = prog.run_backward(ctx.ws, d_src2, d_hy, d_io, ctx.base, grid=runner.grid)
            grads = [None] * ctx.n_params
        else:
            gflat = torch.zeros(ctx.layout["total"], dtype=torch.float32, device=dev)
            d_in = prog.run_backward(ctx.ws, d_src2, d_hy, d_io, gflat.data_ptr(), grid=runner.grid)
            grads = [gflat[o:o + k].view(shape) if o is not None else None for (o, k, shape) in ctx.layout["views"]]
        return (None, d_in if ctx.needs_input_grad[1] else None, None, *grads)


class TailFusedFn(torch.autograd.Function):
    """(x_in, image_tokens) -> (low-res mask logits [n, 4g, 4g], iou [n]): the forward program, the fused bf16 upsampler kernel
    (mp_mask_upsample_fused_bf16, hypernetwork product included), and in the backward that kernel's recomputing twin followed by the
    backward program, which also finishes the upsampler's weight / bias / LayerNorm2d gradients (TailProgram(fused_upsampler=True)).  Five
    launches forward (program, cast, weight pack, upsampler, + the program's counter reset), three backward; no torch arithmetic."""

    @staticmethod
    def forward(ctx, runner, x_in, image_tokens, *params):
        from . import ops
        ctx.set_materialize_grads(False)
        dec = runner.dec
        n, g = x_in.shape[0], dec.grid
        prog, base, layout = runner.program(n, params, x_in.requires_grad)
        ws, src, hyper0, iou4 = prog.run_forward(x_in.detach().contiguous(), image_tokens.detach().contiguous(), grid=runner.grid)
        up = dec.output_upscaling
        w1p, w2p, w1t, w2t = ops.pack_upsampler_weights_all(up[0].weight.detach(), up[3].weight.detach())
        src_bf = ops.cast_to_bf16(src).view(n, g * g, dec.dim)
        eps = float(up[1].eps)
        _, masks = ops.mask_upsample_fused(src_bf, w1p, up[0].bias.detach(), up[1].weight.detach(), up[1].bias.detach(), w2p, up[3].bias.detach(), g, g,
                                           hyper=hyper0, want_up=False, eps=eps)
        ctx.prog, ctx.ws, ctx.runner, ctx.base, ctx.layout, ctx.n_params = prog, ws, runner, base, layout, len(params)
        ctx.ups = (src_bf, w1p, w2p, w1t, w2t, hyper0, eps)
        ctx.params = params
        return masks, iou4[:, 0]

    @staticmethod
    def backward(ctx, d_masks, d_iou):
        from . import ops
        prog, runner = ctx.prog, ctx.runner
        dec = runner.dec
        n, dev, g = prog.n, prog.device, dec.grid
        up = dec.output_upscaling
        src_bf, w1p, w2p, w1t, w2t, hyper0, eps = ctx.ups
        zeros = runner.zeros(n, dev)
        dm = zeros["masks"] if d_masks is None else d_masks.contiguous().float()
        *_, ubuf = ops.mask_upsample_fused_bwd(src_bf, w1p, up[0].bias.detach(), up[1].weight.detach(), up[1].bias.detach(), w2p, up[3].bias.detach(),
                                               hyper0, dm, g, g, eps=eps, w1t=w1t, w2t=w2t, one_buffer=True)
        d_io = zeros["iou"] if d_iou is None else d_iou.contiguous()
        if ctx.base is not None and not _direct_still_valid(ctx.base, ctx.params, runner.direct_offsets(ctx.params, ctx.base, prog)):
            d_in, grads = _detached_grads(lambda b: prog.run_backward_fused(ctx.ws, ubuf, src_bf, d_io, b, grid=runner.grid), ctx.params, prog.grad_offsets, dev)
        elif ctx.base is not None:
            d_in = prog.run_backward_fused(ctx.ws, ubuf, src_bf, d_io, ctx.base, grid=runner.grid)
            grads = [None] * ctx.n_params
        else:
            gflat = torch.zeros(ctx.layout["total"], dtype=torch.float32, device=dev)
            d_in = prog.run_backward_fused(ctx.ws, ubuf, src_bf, d_io, gflat.data_ptr(), grid=runner.grid)
            grads = [gflat[o:o + k].view(shape) if o is not None else None for (o, k, shape) in ctx.layout["views"]]
        return (None, d_in if ctx.needs_input_grad[1] else None, None, *grads)


class TailRunner:
    """Per-MaskDecoder cache of TailPrograms, keyed by what a program is bound to: the prompt count, the addresses of the parameters and of
    their gradients, which of them train."""

    def __init__(self, dec, dense_pe, no_mask_embed, fcs=None, grid=None, fused_upsampler=False):
        import os
        self.dec, self.dense_pe, self.no_mask, self.fcs = dec, dense_pe, no_mask_embed, fcs
        self.fused_upsampler = bool(fused_upsampler)
        self.grid = int(os.environ.get("MP_TAIL_GRID", grid or 256))
        self._cache, self._zeros = {}, {}

    def params(self):
        """Every tensor the program reads as a parameter, in a fixed order (the autograd Function's trailing arguments)."""
        ps = []
        if self.fcs is not None:
            ps += [self.fcs[0].weight, self.fcs[0].bias, self.fcs[1].weight, self.fcs[1].bias]
        d = self.dec
        ps += list(d.transformer.parameters()) + [d.iou_token.weight, d.mask_tokens.weight]
        ps += list(d.output_hypernetworks_mlps[0].parameters()) + list(d.iou_prediction_head.parameters())
        if self.fused_upsampler:
            up = d.output_upscaling
            ps += [up[0].weight, up[0].bias, up[1].weight, up[1].bias, up[3].weight, up[3].bias]
        return ps

    @staticmethod
    def direct_offsets(params, base, prog):
        return prog.grad_offsets

    def zeros(self, n, dev):
        z = self._zeros.get(n)
        if z is None:
            z = {"src2": torch.zeros((2, n, self.dec.grid ** 2, self.dec.dim), dtype=torch.float32, device=dev),
                 "src2_one": torch.zeros((2, n, self.dec.grid ** 2, self.dec.dim), dtype=torch.float32, device=dev),
                 "hy": torch.zeros((n, self.dec.dim // 8), dtype=torch.float32, device=dev), "iou": torch.zeros(n, dtype=torch.float32, device=dev),
                 "masks": torch.zeros((n, 4 * self.dec.grid, 4 * self.dec.grid), dtype=torch.float32, device=dev)}
            self._zeros[n] = z
        return z

    def program(self, n, params, hidden_grad):
        train = [p for p in params if p.requires_grad]
        direct = bool(train) and all(p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32 for p in train)
        if direct:
            st = train[0].grad.untyped_storage().data_ptr()
            direct = all(p.grad.untyped_storage().data_ptr() == st for p in train)
        if direct:
            base = st
            offs = {id(p): p.grad.data_ptr() - base for p in train}
            layout = None
        else:
            base, offs, views, tot = None, {}, [], 0
            for p in params:
                if p.requires_grad:
                    offs[id(p)] = 4 * tot
                    views.append((tot, p.numel(), tuple(p.shape)))
                    tot += -(-p.numel() // 64) * 64
                else:
                    views.append((None, 0, None))
            layout = {"total": max(tot, 64), "views": views}
        key = (n, bool(hidden_grad), tuple(p.data_ptr() for p in params), tuple(sorted(offs.values())), self.dense_pe.data_ptr(), self.no_mask.data_ptr())
        hit = self._cache.get(key)
        if hit is None:
            if len(self._cache) > 16:
                self._cache.clear()
            prog = TailProgram(self.dec, n, self.dense_pe, self.no_mask.detach(), fcs=self.fcs, grad_offsets=offs, hidden_grad=bool(hidden_grad),
                               fused_upsampler=self.fused_upsampler)
            hit = self._cache[key] = prog
        return hit, base, layout

    def __call__(self, x_in, image_tokens):
        """-> (src, hyper0, iou), or (masks, iou) from a fused-upsampler runner."""
        return (TailFusedFn if self.fused_upsampler else TailProgramFn).apply(self, x_in, image_tokens, *self.params())
