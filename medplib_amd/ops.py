"""Thin tensor-level wrappers over the C ABI (include/medplib_hip.h).

torch is used here for device memory and streams only: every function hands `data_ptr()`s, sizes and strides to
libmedplib_hip.so.  Nothing in this module computes with torch ops, and nothing falls back to the CPU."""
import torch

from ._lib import lib

BF16, F32 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_QUICK_GELU, ACT_SILU, ACT_SWIGLU_PAIR = 0, 1, 2, 3, 4, 5
SACT_NONE, SACT_RELU, SACT_GELU, SACT_SIGMOID = 0, 1, 2, 3


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t, dtype, name):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a GPU tensor (medplib_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


def _dt(dtype):
    return BF16 if dtype == torch.bfloat16 else F32


class KernelTimer:
    """Optional HIP-event bracketing of one kernel family (bench.py roofline): events are recorded on the stream the
    kernel is launched on (torch's current stream) and only read back after the timed region.  Every `sample_every`-th launch
    is bracketed (an event pair per launch costs ~3.5 us of queue bubbles: 670 records per step were 2.4 ms of a 91 ms step); the
    stride is coprime to the 4-GEMM period of a decoder layer, so the sample walks through every GEMM shape of the step."""

    def __init__(self, sample_every=11):
        self.records = []          # [work, start_event, end_event, kernel] of the sampled launches
        self.sample_every = max(1, int(sample_every))
        self.launches = 0
        self.total_work = 0.0
        self.per_kernel = {}       # kernel tile (256 / 128) -> [launches, work] over ALL launches
        # expert GEMMs with device-side row counts: the work of a launch is (rows the kernel actually processes) x 2 N K, and those rows
        # are known on the device only.  Every such launch leaves (kernel, counts tensor, slab rows, flop per row, record or None, tag)
        # here; resolve() reads all counts back in ONE transfer after the region (no kernel, no sync inside it) and credits the launches
        # with the KEPT rows — a capacity-dropped token is work the kernel skipped (round-3 review, weak item 6).
        self.pending = []
        self.kept_rows = {}        # tag (decoder layer) -> [kept rows of every expert-GEMM launch with that tag]
        self.batched_tag = None

    def begin(self):
        self.launches += 1
        if self.launches % self.sample_every:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return ev

    def end(self, work, start, rows_dev=None, slab_rows=0, flop_per_row=0.0):
        """work: the launch's algorithmic flop — or, with rows_dev (int32 [E] device row counts, each limited to slab_rows), an upper
        bound that resolve() replaces by flop_per_row x the rows the launch processed."""
        kern = int(lib().raw("mp_gemm_last_kernel")())
        if _TOWER_DEPTH:                 # a frozen tower's launch (throughput tiles: few, fat workgroups by design): its own family (+1000), so
            kern += 1000                 # the dominant decoder kernel's figure is not averaged with launches that idle most of the chip alone
        acc = self.per_kernel.setdefault(kern, [0, 0.0])
        acc[0] += 1
        rec = None
        if start is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            rec = [work, start, ev, kern]
            self.records.append(rec)
        if rows_dev is not None:
            self.pending.append((kern, rows_dev, int(slab_rows), float(flop_per_row), rec, self.batched_tag))
            return
        self.total_work += work
        acc[1] += work

    def resolve(self):
        """Read the device-side row counts of the expert launches (one transfer) and credit them; idempotent."""
        if not self.pending:
            return
        pend, self.pending = self.pending, []
        # one transfer for all launches; the count vectors may differ in length (expert parallelism: [E] on the dense-parallel launches,
        # [experts per rank] on the sharded ones — round-4 advisor), so they travel flattened and are split again by length
        flat = [p[1].to(torch.int32).view(-1) for p in pend]
        counts = torch.cat(flat).cpu().split([f.numel() for f in flat])
        for (kern, _, slab, fpr, rec, tag), c in zip(pend, counts):
            rows = int(c.clamp(max=slab).sum()) if slab > 0 else int(c.sum())
            work = fpr * rows
            self.total_work += work
            self.per_kernel[kern][1] += work
            if rec is not None:
                rec[0] = work
            self.kept_rows.setdefault(tag, []).append(rows)

    def summary(self, kernel=None):
        """-> (sampled work, sampled ms, sampled launches, all launches, all work) after a synchronize; `kernel` = 256 / 128
        restricts everything to the launches that went to that kernel."""
        self.resolve()
        recs = [r for r in self.records if kernel is None or r[3] == kernel]
        total_ms = sum(s.elapsed_time(e) for _, s, e, _ in recs)
        if kernel is None:
            return sum(r[0] for r in recs), total_ms, len(recs), self.launches, self.total_work
        n, w = self.per_kernel.get(kernel, [0, 0.0])
        return sum(r[0] for r in recs), total_ms, len(recs), n, w


GEMM_TIMER = None     # set to a KernelTimer by bench.py
PARAM_EPOCH = 0       # bumped by whatever rewrites parameter VALUES through raw pointers (the fused AdamW step, checkpoint loads): caches of
                      # derived operand images (llama_lora.LoraState.padded) are fresh only within one epoch


# ------------------------------------------------------------------ bf16 trunk ------------------------------------------------
_GEMM_WS = {}


def _ensure_gemm_workspace(device):
    """Register the scratch of the 256x256 GEMM's tail split-K once per device: 96 MiB of fp32
    partials + 384 zeroed arrival tickets.  The library itself never allocates (mp_gemm_set_workspace)."""
    key = torch.device(device).index or 0
    if key not in _GEMM_WS:
        with torch.cuda.device(key):                       # the library files the entry under the current device
            ws = torch.empty(96 << 20, dtype=torch.uint8, device=device)
            tickets = torch.zeros(384, dtype=torch.int32, device=device)
            lib().call("mp_gemm_set_workspace", _p(ws), ws.numel(), _p(tickets), tickets.numel())
        _GEMM_WS[key] = (ws, tickets)


_STREAM_WS = {}
_SIDE_STREAMS = {}


def side_stream(device, name, with_gemm_workspace=False):
    """Process-wide side streams (one per device and role): models share them, so the library's small per-stream workspace table
    is not exhausted by short-lived model instances."""
    key = (torch.device(device).index or 0, name)
    if key not in _SIDE_STREAMS:
        import os
        # MP_SIDE_PRIO (A/B): queue priority of the side streams (-1 = high: a tower's workgroups take freed CUs before the decoder's next tiles, so the
        # towers are done sooner and fewer of the decoder's exact-wave launches start with a CU still held; 0 = default)
        prio = int(os.environ.get("MP_SIDE_PRIO_" + name.upper(), os.environ.get("MP_SIDE_PRIO", "0")))
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device, priority=prio)
    st = _SIDE_STREAMS[key]
    if with_gemm_workspace:
        register_stream_workspace(st)
    return st



def register_stream_workspace(stream):
    """Give `stream` (a torch.cuda.Stream that runs GEMMs concurrently with the default stream) its own split-K scratch."""
    key = stream.cuda_stream
    if key not in _STREAM_WS:
        import ctypes
        ws = torch.empty(96 << 20, dtype=torch.uint8, device=stream.device)
        tickets = torch.zeros(384, dtype=torch.int32, device=stream.device)
        lib().call("mp_gemm_set_stream_workspace", ctypes.c_void_p(key), _p(ws), ws.numel(), _p(tickets), tickets.numel())
        _STREAM_WS[key] = (ws, tickets)


def gemm(a, w, bias=None, residual=None, act=ACT_NONE, out_dtype=torch.bfloat16, out=None, alpha=1.0, m_dev=None):
    """out[M,N] = act(alpha * a[M,K] @ w[N,K]^T + bias) + residual.  a/w bf16 with unit inner stride."""
    _chk(a, torch.bfloat16, "gemm.a"); _chk(w, torch.bfloat16, "gemm.w")
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1 and a.shape[1] == w.shape[1]
    M, K = a.shape
    N = w.shape[0]
    n_out = N // 2 if act == ACT_SWIGLU_PAIR else N
    if out is None:
        out = torch.empty((M, n_out), dtype=out_dtype, device=a.device)
    assert out.stride(1) == 1 and out.shape == (M, n_out)
    if bias is not None:
        _chk(bias, torch.float32, "gemm.bias")
    if residual is not None:
        _chk(residual, torch.bfloat16, "gemm.residual"); assert residual.stride(1) == 1
    _ensure_gemm_workspace(a.device)
    t0 = GEMM_TIMER.begin() if GEMM_TIMER is not None else None
    lib().call("mp_gemm_bf16_nt", _p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), _p(bias), _p(residual),
               residual.stride(0) if residual is not None else 0, M, N, K, act, _dt(out.dtype), float(alpha), _p(m_dev),
               _stream())
    if GEMM_TIMER is not None:
        GEMM_TIMER.end(2.0 * M * N * K, t0)
    return out


def gemm_swiglu_keep(a, w, act_out=None):
    """(act [M, N/2] = silu(gate) * up, gu [M, N] bf16 gate|up in w's interleaved row order) from one GEMM launch (mp_gemm_swiglu_keep_bf16)."""
    _chk(a, torch.bfloat16, "gemm_swiglu_keep.a"); _chk(w, torch.bfloat16, "gemm_swiglu_keep.w")
    M, K = a.shape
    N = w.shape[0]
    assert a.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == K
    act = torch.empty((M, N // 2), dtype=torch.bfloat16, device=a.device) if act_out is None else act_out
    gu = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    assert act.shape == (M, N // 2) and act.stride(1) == 1
    _ensure_gemm_workspace(a.device)
    t0 = GEMM_TIMER.begin() if GEMM_TIMER is not None else None
    lib().call("mp_gemm_swiglu_keep_bf16", _p(a), a.stride(0), _p(w), w.stride(0), _p(act), act.stride(0), _p(gu), gu.stride(0), M, N, K, _stream())
    if GEMM_TIMER is not None:
        GEMM_TIMER.end(2.0 * M * N * K, t0)
    return act, gu


def gemm_tile_policy(mode):
    """Tile choice of this thread's dense bf16 GEMM calls (mp_gemm_tile_policy): 1 = 320x256 tiles where the wave model prefers them
    (default), 0 = 256-row tiles only, 2 = 320-row tiles whenever eligible, 3 = as 2 with tails never split (the frozen towers),
    -1 = process default."""
    global _TILE_POLICY
    lib().call("mp_gemm_tile_policy", int(mode))
    _TILE_POLICY = int(mode)


_TOWER_DEPTH = 0             # > 0 while a frozen tower's forward is being issued (throughput_tiles)
_TILE_POLICY = -1            # what this thread last asked for (the library keeps it thread-local; contexts restore it)


def with_throughput_tiles(fn):
    """Decorator form of `throughput_tiles` for a frozen tower's forward: the tile choice is a property of the MODULE, not of the stream
    arrangement of a particular step, so a step's results do not depend on which streams were switched on (the bit-reproducibility tests
    compare exactly that)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        with throughput_tiles():
            return fn(*args, **kwargs)
    return wrapped


class throughput_tiles:
    """Context for the frozen towers (CLIP tower + projector, SAM-Med2D encoder), which in a training step run on side streams beside the
    decoder (started ahead): their GEMMs take the 320-row tiles whenever they are eligible, however few workgroups that leaves.  Alone, a 60-workgroup launch idles three quarters of the
    chip and the selection model rightly prefers 296 small tiles; beside the decoder's GEMMs (one 147 KB workgroup per CU: nothing
    co-resides) every side workgroup displaces decoder work for exactly its own duration, so what counts is CU x time, not latency —
    CLIP's out_proj is 60 x 39.5 us on 320-row tiles against 296 x 30 us on 128 x 128 ones, fc2 60 x 120 us against 228 split units x
    64 us.  MP_TOWER_THROUGHPUT_TILES=0 switches it off (A/B)."""
    _on = None

    def __enter__(self):
        if throughput_tiles._on is None:
            import os
            throughput_tiles._on = os.environ.get("MP_TOWER_THROUGHPUT_TILES", "1") != "0"
        global _TOWER_DEPTH
        _TOWER_DEPTH += 1
        self._prev = _TILE_POLICY
        if throughput_tiles._on and self._prev == -1:          # an explicit policy of the caller (tests, A/B scripts) wins
            gemm_tile_policy(3)                                # whole tiles: a split tail's units wait, and the decoder's stream owns that
        return self

    def __exit__(self, *exc):
        global _TOWER_DEPTH
        _TOWER_DEPTH -= 1
        if _TILE_POLICY != self._prev:
            gemm_tile_policy(self._prev)
        return False


def gemm_tail_wait(cycles=-1):
    """Shader cycles a unit of the 320-row kernel's split tail waits for its siblings before the tile falls back to its last unit
    (mp_gemm_tail_wait; same bits either way).  cycles >= 0 sets it (0 = never wait), < 0 reads; returns the previous value."""
    return int(lib().raw("mp_gemm_tail_wait")(int(cycles)))


def gemm_last_kernel():
    """320 / 256 / 128: the tile of the kernel the last GEMM call of this thread went to."""
    return int(lib().raw("mp_gemm_last_kernel")())


def gemv_rmsnorm(x, norm_w, eps, w, out_dtype=torch.bfloat16, act=ACT_NONE):
    """act(rmsnorm(x) @ w^T) for M <= 2 rows with the norm folded into the GEMV (bit-identical with rmsnorm + gemv); x [M, K] bf16,
    K % 512 == 0; act NONE or SWIGLU_PAIR (w = the interleaved gate|up matrix, result [M, N / 2])."""
    _chk(x, torch.bfloat16, "gemv_rmsnorm.x"); _chk(w, torch.bfloat16, "gemv_rmsnorm.w"); _chk(norm_w, torch.float32, "gemv_rmsnorm.norm_w")
    M, K = x.shape
    N = w.shape[0]
    assert x.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == K and norm_w.is_contiguous() and norm_w.numel() == K
    out = torch.empty((M, N // 2 if act == ACT_SWIGLU_PAIR else N), dtype=out_dtype, device=x.device)
    lib().call("mp_gemv_rmsnorm_bf16", _p(x), x.stride(0), _p(norm_w), float(eps), _p(w), w.stride(0), _p(out), out.stride(0), M, N, K,
               int(act), _dt(out_dtype), _stream())
    return out


def gemv_rmsnorm_rope_append(x, norm_w, eps, w_qkv, cos_t, sin_t, cache_k, cache_v, pos_dev, heads, head_dim):
    """One decode-step launch for input_layernorm -> q|k|v projection -> RoPE at pos_dev[0] -> KV-cache append (bit-identical with
    rmsnorm + gemv + decode_rope_append).  Returns qkv [M, 3*H*D] of which only the q third is written (the rotated q)."""
    _chk(x, torch.bfloat16, "gemv_rmsnorm_rope.x"); _chk(w_qkv, torch.bfloat16, "gemv_rmsnorm_rope.w"); _chk(pos_dev, torch.int32, "gemv_rmsnorm_rope.pos")
    M, K = x.shape
    assert w_qkv.shape == (3 * heads * head_dim, K) and x.stride(1) == 1 and w_qkv.stride(1) == 1
    assert cache_k.stride(3) == 1 and cache_k.stride(2) == head_dim and cache_v.stride() == cache_k.stride()
    qkv = torch.empty((M, 3 * heads * head_dim), dtype=torch.bfloat16, device=x.device)
    lib().call("mp_gemv_rmsnorm_rope_append_bf16", _p(x), x.stride(0), _p(norm_w), float(eps), _p(w_qkv), w_qkv.stride(0), _p(qkv), qkv.stride(0),
               _p(cos_t), _p(sin_t), _p(cache_k), _p(cache_v), _p(pos_dev), M, heads, head_dim, K, cache_k.stride(0), cache_k.stride(1), _stream())
    return qkv


def gemv_rmsnorm_ok(M, K, head_dim=None, swiglu_n=None):
    """Whether the decode step may use the norm-folded GEMV launches (mp_gemv_rmsnorm_bf16 / mp_gemv_rmsnorm_rope_append_bf16) — ALL of their
    preconditions, so that a configuration outside them falls back to rmsnorm + gemv + decode_rope_append instead of raising from C
    (round-3 advisor): M <= 2 rows, K / 512 a power of two up to 16 (the normed rows live in 4 * M * K * 2 bytes of static LDS: 128 KiB at
    K = 8192, M = 2), and — when the caller folds them — head_dim % 16 == 0 for the RoPE / cache-append tail and 2 * ff % 64 == 0 for the
    SwiGLU-paired gate|up form."""
    if not (M <= 2 and K % 512 == 0 and (K // 512) in (1, 2, 4, 8, 16) and 4 * M * K * 2 <= 160 * 1024):
        return False
    if head_dim is not None and head_dim % 16:
        return False
    if swiglu_n is not None and swiglu_n % 64:
        return False
    return True


def gemv(x, w, bias=None, residual=None, act=ACT_NONE, out_dtype=torch.bfloat16, alpha=1.0, w_index=None, row_scale=None, row_keep=None):
    """x [M <= 8, K] bf16; w [N, K] bf16 or, with w_index (int32 [M] device), [E, N, K] with row m using w[w_index[m]].
    Decode-step projections (HBM-bound weight stream)."""
    _chk(x, torch.bfloat16, "gemv.x"); _chk(w, torch.bfloat16, "gemv.w")
    M, K = x.shape
    N = w.shape[-2]
    assert x.stride(1) == 1 and w.stride(-1) == 1
    out = torch.empty((M, N // 2 if act == ACT_SWIGLU_PAIR else N), dtype=out_dtype, device=x.device)
    lib().call("mp_gemv_bf16", _p(x), x.stride(0), _p(w), w.stride(-2), w.stride(0) if w.dim() == 3 else 0, _p(out), out.stride(0), _p(bias),
               _p(residual), residual.stride(0) if residual is not None else 0, _p(w_index), _p(row_scale), _p(row_keep), M, N, K, act,
               _dt(out_dtype), float(alpha), _stream())
    return out


def gemm_batched(a, w, out, m_dev=None, bias=None, act=ACT_NONE):
    """a [E,M,K], w [E,N,K], out [E,M,N] (bf16 or f32); m_dev int32 [E] device row counts."""
    _chk(a, torch.bfloat16, "gemm_batched.a"); _chk(w, torch.bfloat16, "gemm_batched.w")
    E, M, K = a.shape
    N = w.shape[1]
    assert a.stride(2) == 1 and w.stride(2) == 1 and out.stride(2) == 1
    _ensure_gemm_workspace(a.device)
    t0 = GEMM_TIMER.begin() if GEMM_TIMER is not None else None
    lib().call("mp_gemm_bf16_nt_batched", _p(a), a.stride(1), a.stride(0), _p(w), w.stride(1), w.stride(0), _p(out),
               out.stride(1), out.stride(0), _p(bias), bias.stride(0) if bias is not None else 0, E, M, N, K, act,
               _dt(out.dtype), _p(m_dev), _stream())
    if GEMM_TIMER is not None:
        # algorithmic rows = the rows the kernel processes: the device-side counts (read back after the region), else every slab row
        GEMM_TIMER.end(2.0 * E * M * N * K, t0, rows_dev=m_dev, slab_rows=M, flop_per_row=2.0 * N * K)
    return out


def gemm_batched_res(a, w, residual, out, m_dev=None):
    """out[e] = bf16(a[e] @ w[e]^T) + residual[e]; a [E,M,K], w [E,N,K], residual / out [E,M,N] bf16."""
    E, M, K = a.shape
    N = w.shape[1]
    assert a.stride(2) == 1 and w.stride(2) == 1 and out.stride(2) == 1 and residual.stride(2) == 1
    _ensure_gemm_workspace(a.device)
    lib().call("mp_gemm_bf16_nt_batched_res", _p(a), a.stride(1), a.stride(0), _p(w), w.stride(1), w.stride(0), _p(out), out.stride(1),
               out.stride(0), _p(residual), residual.stride(1), residual.stride(0), E, M, N, K, _p(m_dev), _stream())
    return out


def attention(q, k, v, out=None, causal=False, key_valid=None, rel_h=None, rel_w=None, scale=None, variant=0, sk_dev=None):
    """q,k,v: [B,S,H,D] bf16 views (stride(3)==1, stride(2)==D); returns [B,Sq,H*D] bf16."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, torch.bfloat16, "attention." + n)
        assert t.dim() == 4 and t.stride(3) == 1 and t.stride(2) == t.shape[3]
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    if out is None:
        out = torch.empty((B, Sq, H * D), dtype=torch.bfloat16, device=q.device)
    assert out.stride(2) == 1
    if scale is None:
        scale = D ** -0.5
    kh = kw = 0
    if rel_h is not None:
        _chk(rel_h, torch.float32, "attention.rel_h"); _chk(rel_w, torch.float32, "attention.rel_w")
        assert rel_h.is_contiguous() and rel_w.is_contiguous()
        kh, kw = rel_h.shape[-1], rel_w.shape[-1]
    if key_valid is not None:
        _chk(key_valid, torch.uint8, "attention.key_valid"); assert key_valid.is_contiguous()
    if Sq == 1:
        _ensure_gemm_workspace(q.device)          # the decode kernel splits a head's keys over workgroups through this scratch
    lib().call("mp_attention_fwd_bf16", _p(q), q.stride(0), q.stride(1), _p(k), k.stride(0), k.stride(1), _p(v), v.stride(0),
               v.stride(1), _p(out), out.stride(0), out.stride(1), _p(key_valid), _p(rel_h), _p(rel_w), kh, kw, B, H, Sq, Sk,
               D, int(causal), float(scale), variant, _p(sk_dev), _stream())
    return out


def attention_fwd_lse(q, k, v, causal=True, key_valid=None, scale=None, out=None):
    """Attention forward that also returns the row log-sum-exp (log2 domain) for the backward.  q,k,v: [B,S,H,D] bf16 views.
    -> (out [B,Sq,H*D] bf16, lse2 [B*H, Sq] fp32)."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, torch.bfloat16, "attention_fwd_lse." + n)
        assert t.dim() == 4 and t.stride(3) == 1 and t.stride(2) == t.shape[3]
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    if out is None:
        out = torch.empty((B, Sq, H * D), dtype=torch.bfloat16, device=q.device)
    assert out.shape == (B, Sq, H * D) and out.stride(2) == 1
    lse2 = torch.empty((B * H, Sq), dtype=torch.float32, device=q.device)
    if key_valid is not None:
        _chk(key_valid, torch.uint8, "attention_fwd_lse.key_valid"); assert key_valid.is_contiguous()
    lib().call("mp_attention_fwd_lse_bf16", _p(q), q.stride(0), q.stride(1), _p(k), k.stride(0), k.stride(1), _p(v), v.stride(0), v.stride(1),
               _p(out), out.stride(0), out.stride(1), _p(key_valid), B, H, Sq, Sk, D, int(bool(causal)),
               float(D ** -0.5 if scale is None else scale), _p(lse2), _stream())
    return out, lse2


def attention_bwd(q, k, v, out, d_out, lse2, causal=True, key_valid=None, scale=None, fused_delta=True, rope=None):
    """Backward of softmax(scale q k^T + mask) v.  q,k,v [B,S,H,D] bf16 views, out / d_out [B,Sq,H*D] bf16, lse2 from attention_fwd_lse.
    fused_delta: delta = rowsum(dO o O) inside the dQ kernel (default) instead of the separate mp_attention_delta_bf16 pass.
    rope = (cos, sin) fp32 [positions, D / 2]: dq and dk come back rotated by them (pass the negated sin for the transpose of RoPE).
    -> (dq, dk, dv, dqkv): [B,S,H,D] views of one [B,S,3,H,D] buffer (the layout of the fused qkv projection output) and the buffer."""
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    assert Sq == Sk, "self-attention backward (the fused dqkv buffer holds one row per position)"
    _chk(d_out, torch.bfloat16, "attention_bwd.d_out"); _chk(out, torch.bfloat16, "attention_bwd.out")
    assert d_out.stride(2) == 1 and out.stride(2) == 1
    delta = torch.empty((B * H, Sq), dtype=torch.float32, device=q.device)
    dqkv = padded_rows(B * Sq, 3 * H * D, q.device).unflatten(0, (B, Sq)).unflatten(2, (3, H, D))
    dq, dk, dv = dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]
    sc = float(D ** -0.5 if scale is None else scale)
    if fused_delta and out.stride(1) % 8 == 0:
        lib().call("mp_attention_bwd_fused_bf16", _p(q), q.stride(0), q.stride(1), _p(k), k.stride(0), k.stride(1), _p(v), v.stride(0),
                   v.stride(1), _p(out), out.stride(0), out.stride(1), _p(d_out), d_out.stride(0), d_out.stride(1), _p(lse2), _p(delta),
                   _p(dq), dq.stride(0), dq.stride(1), _p(dk), dk.stride(0), dk.stride(1), _p(dv), dv.stride(0), dv.stride(1),
                   _p(key_valid), B, H, Sq, Sk, D, int(bool(causal)), sc, _p(rope[0]) if rope else None, _p(rope[1]) if rope else None, _stream())
        return dq, dk, dv, dqkv
    assert rope is None, "attention_bwd: the rotated store needs the fused-delta form"
    lib().call("mp_attention_delta_bf16", _p(out), out.stride(0), out.stride(1), _p(d_out), d_out.stride(0), d_out.stride(1), _p(delta),
               B, H, Sq, D, _stream())
    lib().call("mp_attention_bwd_bf16", _p(q), q.stride(0), q.stride(1), _p(k), k.stride(0), k.stride(1), _p(v), v.stride(0), v.stride(1),
               _p(d_out), d_out.stride(0), d_out.stride(1), _p(lse2), _p(delta), _p(dq), dq.stride(0), dq.stride(1), _p(dk), dk.stride(0),
               dk.stride(1), _p(dv), dv.stride(0), dv.stride(1), _p(key_valid), B, H, Sq, Sk, D, int(bool(causal)), sc, _stream())
    return dq, dk, dv, dqkv


# ------------------------------------------------------------------ decoder backward pieces (LoRA training) --------------------
def rmsnorm_bwd(x, w, dy, eps, add=None, want_wgrad=False):
    """dx of LlamaRMSNorm, optionally + add (the residual-stream gradient).  x, dy [T, d] bf16.  want_wgrad: -> (dx, dw [d] fp32)."""
    _chk(x, torch.bfloat16, "rmsnorm_bwd.x"); _chk(dy, torch.bfloat16, "rmsnorm_bwd.dy")
    T, d = x.shape
    out = torch.empty((T, d), dtype=torch.bfloat16, device=x.device)
    rs = torch.empty(T, dtype=torch.float32, device=x.device) if want_wgrad else None
    lib().call("mp_rmsnorm_bwd_bf16", _p(x), x.stride(0), _p(w), _p(dy), dy.stride(0), _p(add), add.stride(0) if add is not None else 0,
               _p(out), out.stride(0), T, d, float(eps), _p(rs), _stream())
    if not want_wgrad:
        return out
    dw = torch.empty(d, dtype=torch.float32, device=x.device)
    partial = torch.empty(((T + 255) // 256) * d, dtype=torch.float32, device=x.device)
    lib().call("mp_rmsnorm_wgrad_f32", _p(x), x.stride(0), _p(dy), dy.stride(0), _p(rs), _p(dw), _p(partial), partial.numel(), T, d, _stream())
    return out, dw


def rmsnorm_bwd_up(x, w, dy, eps, dt, AT, R, p=0.0, seed=0, add=None, keep_bits=None):
    """rmsnorm_bwd(x, w, lora_up_add(dt, AT, dy, R, p, seed), eps, add) in one pass (mp_rmsnorm_bwd_up_bf16; dy is not modified), bit-identical."""
    _chk(x, torch.bfloat16, "rmsnorm_bwd_up.x"); _chk(dy, torch.bfloat16, "rmsnorm_bwd_up.dy"); _chk(dt, torch.bfloat16, "rmsnorm_bwd_up.dt")
    T, d = x.shape
    assert AT.shape == (d, 64) and AT.is_contiguous() and dt.shape[0] == T and dt.stride(1) == 1 and dy.shape == (T, d)
    out = torch.empty((T, d), dtype=torch.bfloat16, device=x.device)
    kb = keep_bits if p > 0 else None
    lib().call("mp_rmsnorm_bwd_up_bf16", _p(x), x.stride(0), _p(w), _p(dy), dy.stride(0), _p(add), add.stride(0) if add is not None else 0, _p(out),
               out.stride(0), T, d, float(eps), _p(dt), dt.stride(0), _p(AT), int(R), float(p), int(seed), _p(kb), kb.stride(0) if kb is not None else 0,
               _stream())
    return out


def swiglu_pair_fwd(gu, out=None, counts=None, cap=0):
    """counts / cap: the rows are capacity slabs [E * cap, .]; only the first counts[e] rows of slab e are processed (the rest is left alone)."""
    T, ff2 = gu.shape
    act = torch.empty((T, ff2 // 2), dtype=torch.bfloat16, device=gu.device) if out is None else out
    assert gu.is_contiguous() and act.shape == (T, ff2 // 2) and act.stride(1) == 1
    lib().call("mp_swiglu_pair_fwd_bf16", _p(gu), _p(act), act.stride(0), T, ff2 // 2, _p(counts), int(cap), _stream())
    return act


def swiglu_pair_bwd(gu, dact, counts=None, cap=0):
    T, ff2 = gu.shape
    dgu = torch.empty_like(gu)
    lib().call("mp_swiglu_pair_bwd_bf16", _p(gu), _p(dact), _p(dgu), T, ff2 // 2, _p(counts), int(cap), _stream())
    return dgu


class SkinnyPartial:
    """The chunk partials of one tn_skinny product, not yet summed: [chunks, N * R] fp32 + the scale the sum takes.  lora_grad_unpack_partials
    consumes it directly; .finish() is the reduce (same values as tn_skinny(..., reduce=True))."""
    __slots__ = ("partial", "chunks", "scale", "N", "R")

    def __init__(self, partial, chunks, scale, N, R):
        self.partial, self.chunks, self.scale, self.N, self.R = partial, chunks, scale, N, R

    def finish(self):
        return colsum_scaled(self.partial.view(self.chunks, self.N * self.R), self.scale).view(self.N, self.R)


def colsum_scaled(x2d, scale):
    out = colsum_f32(x2d)
    return out if scale == 1.0 else scale_f32_(out, scale)


def tn_skinny(x, g, R, scale=1.0, p=0.0, seed=0, rows_dev=None, reduce=True, keep_bits=None):
    """out[n, j] = scale * sum_t drop(x)[t, n] * g[t, j], j < R: fp32 [N, R].  x [T, N] bf16, g [T, >= 16 * ceil(R / 16)] bf16; p > 0: x is the
    undropped tensor and the lora_dropout mask (mp_dropout_bf16 over the contiguous [T, N]) is applied on the way.  rows_dev: int32 device
    scalar, only the first min(T, rows_dev) rows count (an expert's routed rows on its capacity slab)."""
    _chk(x, torch.bfloat16, "tn_skinny.x"); _chk(g, torch.bfloat16, "tn_skinny.g")
    T, N = x.shape
    if R > 32:                      # wider than the kernel's register budget: two passes over column halves of g
        return torch.cat([tn_skinny(x, g[:, j:j + 32], 32, scale, p, seed, rows_dev, keep_bits=keep_bits) for j in range(0, R, 32)], dim=1)
    need = (R + 15) // 16 * 16                                  # the MFMA kernel reads g in 16-column groups
    if g.shape[1] < need:
        gp = torch.zeros((T, need), dtype=torch.bfloat16, device=g.device)
        gp[:, :g.shape[1]] = g
        g = gp
    chunks = (T + 255) // 256
    partial = torch.empty(chunks * N * R, dtype=torch.float32, device=x.device)
    out = torch.empty((N, R), dtype=torch.float32, device=x.device) if reduce else None
    kb = keep_bits if p > 0 else None
    lib().call("mp_tn_skinny_f32", _p(x), x.stride(0), _p(g), g.stride(0), _p(out), _p(partial), partial.numel(), T, N, int(R), float(scale),
               float(p), int(seed), _p(rows_dev), _p(kb), kb.stride(0) if kb is not None else 0, _stream())
    return out if reduce else SkinnyPartial(partial, chunks, float(scale), N, int(R))


def tn_skinny_down(x, g, Bt, R, scale=1.0, alpha=1.0, reduce=True):
    """(tn_skinny(x, g, R, scale), lora_down(x, Bt, ., R, alpha=alpha)) from ONE pass over x [T, N] (mp_tn_skinny_down_f32): the weight gradient
    dB = scale * x^T g [N, R] fp32 (or its chunk partials) and dt [T, 64] bf16 = alpha * x Bt^T.  Bt [>= 16 * ceil(R / 16), N] bf16."""
    _chk(x, torch.bfloat16, "tn_skinny_down.x"); _chk(g, torch.bfloat16, "tn_skinny_down.g"); _chk(Bt, torch.bfloat16, "tn_skinny_down.Bt")
    T, N = x.shape
    rg = (R + 15) // 16
    assert R <= 32 and g.shape[1] >= 16 * rg and Bt.shape[1] == N and Bt.shape[0] >= 16 * rg and x.stride(1) == 1 and Bt.stride(1) == 1 and g.stride(1) == 1
    chunks, blocks = (T + 255) // 256, (N + 255) // 256
    partial = torch.empty(chunks * N * R, dtype=torch.float32, device=x.device)
    dtp = torch.empty(blocks * T * 16 * rg, dtype=torch.float32, device=x.device)
    dt = torch.empty((T, 64), dtype=torch.bfloat16, device=x.device)
    out = torch.empty((N, R), dtype=torch.float32, device=x.device) if reduce else None
    lib().call("mp_tn_skinny_down_f32", _p(x), x.stride(0), _p(g), g.stride(0), _p(out), _p(partial), partial.numel(), _p(Bt), Bt.stride(0), _p(dt), dt.stride(0),
               _p(dtp), dtp.numel(), T, N, int(R), float(scale), float(alpha), _stream())
    return (out if reduce else SkinnyPartial(partial, chunks, float(scale), N, int(R))), dt


def swiglu_bwd_skinny(dt_down, AT_down, dact, gu, R_down, p, seed, t_gu, Bt_gu, R_gu, scale=1.0, alpha=1.0, reduce=True, keep_bits=None):
    """lora_up_add_swiglu_bwd(dt_down, AT_down, dact, gu, R_down, p, seed) and tn_skinny_down(that, t_gu, Bt_gu, R_gu, scale, alpha) in one kernel
    (mp_swiglu_bwd_skinny_f32), the same bits: -> (d gate|up [T, 2 ff] bf16, dB_gu (or its chunk partials), dt_gu [T, 64] bf16)."""
    _chk(dact, torch.bfloat16, "swiglu_bwd_skinny.dact"); _chk(gu, torch.bfloat16, "swiglu_bwd_skinny.gu"); _chk(t_gu, torch.bfloat16, "swiglu_bwd_skinny.t")
    T, ff = dact.shape
    N, rg = 2 * ff, (R_gu + 15) // 16
    assert AT_down.shape == (ff, 64) and AT_down.is_contiguous() and gu.shape == (T, N) and gu.is_contiguous() and dact.stride(1) == 1
    assert R_gu <= 32 and t_gu.shape[1] >= 16 * rg and Bt_gu.shape[1] == N and Bt_gu.shape[0] >= 16 * rg and Bt_gu.stride(1) == 1 and t_gu.stride(1) == 1
    chunks, blocks = (T + 255) // 256, (N + 255) // 256
    dgu = torch.empty_like(gu)
    partial = torch.empty(chunks * N * R_gu, dtype=torch.float32, device=dact.device)
    dtp = torch.empty(blocks * T * 16 * rg, dtype=torch.float32, device=dact.device)
    dt = torch.empty((T, 64), dtype=torch.bfloat16, device=dact.device)
    out = torch.empty((N, R_gu), dtype=torch.float32, device=dact.device) if reduce else None
    kb = keep_bits if p > 0 else None
    lib().call("mp_swiglu_bwd_skinny_f32", _p(dact), dact.stride(0), _p(gu), _p(dgu), _p(dt_down), dt_down.stride(0), _p(AT_down), int(R_down), float(p),
               int(seed), _p(kb), kb.stride(0) if kb is not None else 0, _p(t_gu), t_gu.stride(0), _p(out), _p(partial), partial.numel(), _p(Bt_gu),
               Bt_gu.stride(0), _p(dt), dt.stride(0), _p(dtp), dtp.numel(), T, ff, int(R_gu), float(scale), float(alpha), _stream())
    return dgu, (out if reduce else SkinnyPartial(partial, chunks, float(scale), N, int(R_gu))), dt


def lora_grad_unpack_partials(dB, dAT, rows, k0, gB, gA):
    """lora_grad_unpack from two SkinnyPartial (mp_lora_grad_unpack_partials_f32): the chunk sums happen in the unpack itself."""
    fout, r = gB.shape
    fin = gA.shape[1]
    assert dB.R == dAT.R and dAT.N == fin and gB.is_contiguous() and gA.is_contiguous() and rows.numel() == fout
    lib().call("mp_lora_grad_unpack_partials_f32", _p(dB.partial), _p(dAT.partial), dB.chunks, dAT.chunks, dB.scale, dAT.scale, _p(rows), dB.R, int(k0), r, fin,
               fout, dB.N, _p(gB), _p(gA), _stream())


def ce_rows_bwd(logits, labels, gscale, gconst, ldo):
    """bf16 [n, ldo] = gconst * gscale[0] * (softmax(logits) - onehot(labels)), zero beyond V."""
    n, V = logits.shape
    out = torch.empty((n, ldo), dtype=torch.bfloat16, device=logits.device)
    lib().call("mp_ce_rows_bwd", _p(logits), logits.stride(0), _p(labels), _p(gscale), float(gconst), _p(out), ldo, n, V, _stream())
    return out


def scatter_rows_f32_bf16(g, rows, T):
    """zeros [T, d] bf16 with row rows[i] = bf16(g[i])."""
    _chk(g, torch.float32, "scatter_rows.g")
    n, d = g.shape
    out = torch.zeros((T, d), dtype=torch.bfloat16, device=g.device)
    lib().call("mp_scatter_rows_f32_bf16", _p(g.contiguous()), _p(rows), _p(out), n, d, _stream())
    return out


def lora_pack(a, b, rows, A, AT, B, BT, k0, bscale=1.0, Bx=None, xscale=1.0):
    """fp32 adapter (a [r, fin], b [fout, r]) -> its slices of the padded bf16 operands (both orientations); B is stored * bscale.  Bx
    (optional): a [W, 64] view (any row stride) that receives b * xscale as well -- the K-extension columns of an extended weight."""
    r, fin = a.shape
    fout = b.shape[0]
    lib().call("mp_lora_pack", _p(a), _p(b), _p(rows), _p(A), _p(AT), _p(B), _p(BT), r, fin, fout, int(k0), B.shape[0], float(bscale),
               _p(Bx), Bx.stride(0) if Bx is not None else 0, float(xscale), _stream())


def lora_grad_unpack(dB, dAT, rows, k0, gB, gA):
    """gB [fout, r] += dB[rows, k0:k0 + r], gA [r, fin] += dAT[:, k0:k0 + r].T in one launch (mp_lora_grad_unpack_f32); all fp32, contiguous."""
    for t_, n_ in ((dB, "dB"), (dAT, "dAT"), (gB, "gB"), (gA, "gA")):
        _chk(t_, torch.float32, "lora_grad_unpack." + n_); assert t_.is_contiguous()
    fout, r = gB.shape
    fin = gA.shape[1]
    assert gA.shape[0] == r and dAT.shape[0] == fin and dB.shape[1] == dAT.shape[1] and rows.numel() == fout
    lib().call("mp_lora_grad_unpack_f32", _p(dB), _p(dAT), _p(rows), dB.shape[1], int(k0), r, fin, fout, _p(gB), _p(gA), _stream())


def keep_bits_for(x):
    """Room for the lora_dropout mask of x [T, K] as bytes [T, K / 8] (mp_lora_down_bf16 writes it, the backward kernels read it), or None when
    the kernel that writes it does not take this shape (K % 256 != 0: the mask is then regenerated from the seed as before)."""
    T, K = x.shape
    return torch.empty((T, K // 8), dtype=torch.uint8, device=x.device) if K % 256 == 0 else None


def lora_down(x, A, t, R, p=0.0, seed=0, xd=None, alpha=1.0, rows_dev=None, keep_bits=None):
    """t[:, :64] = bf16(dropout(x) @ A[:R]^T) (zeros beyond R), the dropped x into xd when given (mp_lora_down_bf16).  x [T, K] bf16 (any row
    stride), A [>= 16 * ceil(R / 16), K] bf16, t a [T, 64] view (typically columns K.. of x's own row-padded buffer)."""
    _chk(x, torch.bfloat16, "lora_down.x"); _chk(A, torch.bfloat16, "lora_down.A"); _chk(t, torch.bfloat16, "lora_down.t")
    T, K = x.shape
    assert x.stride(1) == 1 and A.stride(1) == 1 and t.stride(1) == 1 and t.shape == (T, 64) and A.shape[1] == K and A.shape[0] >= (R + 15) // 16 * 16
    if xd is not None:
        assert xd.shape == (T, K) and xd.stride(1) == 1
    partial = torch.empty(8 * T * 16 * ((R + 15) // 16), dtype=torch.float32, device=x.device) if K % 256 == 0 else None
    lib().call("mp_lora_down_bf16", _p(x), x.stride(0), _p(A), A.stride(0), _p(t), t.stride(0), _p(xd), xd.stride(0) if xd is not None else 0,
               T, K, int(R), float(p), int(seed), float(alpha), _p(rows_dev), _p(partial), partial.numel() if partial is not None else 0,
               _p(keep_bits) if p > 0 else None, keep_bits.stride(0) if (keep_bits is not None and p > 0) else 0, _stream())
    return t


def lora_up_add(dt, AT, dx, R, p=0.0, seed=0, out=None, rows_dev=None, keep_bits=None):
    """dx + dropout(bf16(dt @ AT^T)) with mp_dropout_bf16's mask (mp_lora_up_add_bf16): dt [T, >= R] bf16, AT [K, 64] bf16 (A^T, padded),
    dx [T, K] bf16; in place on dx unless `out` is given."""
    _chk(dt, torch.bfloat16, "lora_up_add.dt"); _chk(AT, torch.bfloat16, "lora_up_add.AT"); _chk(dx, torch.bfloat16, "lora_up_add.dx")
    T, K = dx.shape
    assert AT.shape == (K, 64) and AT.is_contiguous() and dt.shape[0] == T and dt.stride(1) == 1 and dx.stride(1) == 1
    out = dx if out is None else out
    kb = keep_bits if p > 0 else None
    lib().call("mp_lora_up_add_bf16", _p(dt), dt.stride(0), _p(AT), _p(dx), dx.stride(0), _p(out), out.stride(0), T, K, int(R), float(p), int(seed),
               _p(rows_dev), _p(kb), kb.stride(0) if kb is not None else 0, _stream())
    return out


def lora_up_add_swiglu_bwd(dt, AT, dact, gu, R, p=0.0, seed=0, keep_bits=None):
    """swiglu_pair_bwd(gu, lora_up_add(dt, AT, dact)) in one pass (mp_lora_up_add_swiglu_bwd_bf16), bit-identical with the two kernels:
    dt [T, >= R] bf16, AT [ff, 64] bf16, dact [T, ff] bf16 (not modified), gu [T, 2 ff] bf16 -> d gate|up [T, 2 ff] bf16."""
    _chk(dt, torch.bfloat16, "lora_up_add_swiglu_bwd.dt"); _chk(dact, torch.bfloat16, "lora_up_add_swiglu_bwd.dact"); _chk(gu, torch.bfloat16, "lora_up_add_swiglu_bwd.gu")
    T, ff = dact.shape
    assert AT.shape == (ff, 64) and AT.is_contiguous() and dt.shape[0] == T and dt.stride(1) == 1 and dact.stride(1) == 1
    assert gu.shape == (T, 2 * ff) and gu.is_contiguous()
    dgu = torch.empty_like(gu)
    kb = keep_bits if p > 0 else None
    lib().call("mp_lora_up_add_swiglu_bwd_bf16", _p(dt), dt.stride(0), _p(AT), _p(dact), dact.stride(0), _p(gu), _p(dgu), T, ff, int(R), float(p), int(seed),
               _p(kb), kb.stride(0) if kb is not None else 0, _stream())
    return dgu


def moe_combine_bwd(dout, y, expert, slot, weight, capacity, top_k=1):
    """-> (d_y [E, cap, d] bf16 (zeros where no token sits), d_w [top_k * T] fp32)."""
    T, d = dout.shape
    dy = torch.zeros_like(y)
    dw = torch.empty(T * top_k, dtype=torch.float32, device=dout.device)
    lib().call("mp_moe_combine_bwd_bf16", _p(dout), _p(y), _p(expert), _p(slot), _p(weight), _p(dy), _p(dw), T, d, int(capacity), int(top_k),
               _stream())
    return dy, dw


def moe_gate_bwd(gates, expert, slot, dw, first_counts, c_aux, aux_coef, top_k=1):
    T, E = gates.shape
    dl = torch.empty((T, E), dtype=torch.float32, device=gates.device)
    lib().call("mp_moe_gate_bwd_f32", _p(gates), _p(expert), _p(slot), _p(dw), _p(first_counts), _p(c_aux), float(aux_coef), _p(dl), T, E,
               int(top_k), _stream())
    return dl


def moe_gate_dgrad_(dlogits, wg, dx):
    """dx += dlogits @ wg, in place on the bf16 [T, d] gradient."""
    T, d = dx.shape
    lib().call("mp_moe_gate_dgrad_bf16", _p(dlogits), _p(wg), _p(dx), T, d, wg.shape[0], _stream())
    return dx


def embed_grad(g, rows_sorted, seg, ids, vocab):
    """fp32 [vocab, d] gradient of the embedding table from the bf16 row gradients g [T, d] (segments of rows per token id)."""
    T, d = g.shape
    out = torch.zeros((vocab, d), dtype=torch.float32, device=g.device)
    lib().call("mp_embed_grad_f32", _p(g), _p(rows_sorted), _p(seg), _p(ids), _p(out), ids.numel(), d, _stream())
    return out


def gelu_fwd_bf16(x):
    y = torch.empty_like(x)
    lib().call("mp_gelu_fwd_bf16", _p(x), _p(y), x.numel(), _stream())
    return y


def gelu_bwd_bf16(x, dy):
    dx = torch.empty_like(x)
    lib().call("mp_gelu_bwd_bf16", _p(x), _p(dy), _p(dx), x.numel(), _stream())
    return dx


def region_point_mean_bwd(xy, offsets, map_index, dout, n_maps, h, w):
    """d(feature maps) [n_maps, h*w, C] bf16 of region_point_mean from dout [n_masks, C] bf16."""
    n_masks, C = dout.shape
    dfmap = torch.empty((n_maps, h * w, C), dtype=torch.bfloat16, device=dout.device)
    wt = torch.empty(max(n_masks, 1) * h * w, dtype=torch.float32, device=dout.device)
    lib().call("mp_region_point_mean_bwd_bf16", _p(xy), _p(offsets), _p(map_index), _p(dout.contiguous()), _p(dfmap), _p(wt), n_maps, n_masks,
               h, w, C, _stream())
    return dfmap


def conv3x3s2_c1_pre(img, w, b):
    """Conv2d(1, CO, k3, s2, p1) without the GELU: img [n, H, W] (bf16 / f32) -> pre-activation [n, OH, OW, CO] bf16."""
    n, H, W = img.shape
    CO = w.shape[0]
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    out = torch.empty((n, OH, OW, CO), dtype=torch.bfloat16, device=img.device)
    lib().call("mp_conv3x3s2_c1_pre_bf16", _p(img), _dt(img.dtype), _p(w), _p(b), _p(out), n, H, W, CO, _stream())
    return out


def conv3x3s2_c1_wgrad(img, dpre):
    n, H, W = img.shape
    CO = dpre.shape[-1]
    dw = torch.empty((CO, 9), dtype=torch.float32, device=img.device); db = torch.empty(CO, dtype=torch.float32, device=img.device)
    lib().call("mp_conv3x3s2_c1_wgrad_f32", _p(img), _dt(img.dtype), _p(dpre.contiguous()), _p(dw), _p(db), n, H, W, CO, _stream())
    return dw, db


def adaptive_avgpool_tokens_bwd(dout, len_in):
    n, len_out, C = dout.shape
    dx = torch.empty((n, len_in, C), dtype=torch.bfloat16, device=dout.device)
    lib().call("mp_adaptive_avgpool_tokens_bwd_bf16", _p(dout.contiguous()), _p(dx), n, len_in, len_out, C, _stream())
    return dx


def col2im_k3s2p1(dcols, n, H, W, C):
    dx = torch.empty((n, H, W, C), dtype=torch.bfloat16, device=dcols.device)
    lib().call("mp_col2im_k3s2p1_bf16", _p(dcols.contiguous()), _p(dx), n, H, W, C, _stream())
    return dx


def dropout_bf16(x, p, seed):
    y = torch.empty_like(x)
    lib().call("mp_dropout_bf16", _p(x), _p(y), x.numel(), float(p), int(seed), _stream())
    return y


def decode_rope_append(qkv, cos_t, sin_t, cache_k, cache_v, pos_dev, heads, head_dim):
    """qkv [B, 3*H*D] (one new token per sequence): q rotated in place, rotated k / v written to the caches [B, max_len, H, D] at
    position pos_dev[0] (int32 on the device)."""
    _chk(qkv, torch.bfloat16, "decode_rope_append.qkv"); _chk(pos_dev, torch.int32, "decode_rope_append.pos")
    assert qkv.dim() == 2 and qkv.stride(1) == 1 and cache_k.stride(3) == 1 and cache_k.stride(2) == head_dim
    lib().call("mp_decode_rope_append_bf16", _p(qkv), qkv.stride(0), _p(cos_t), _p(sin_t), _p(cache_k), _p(cache_v), _p(pos_dev), qkv.shape[0],
               heads, head_dim, cache_k.stride(0), cache_k.stride(1), _stream())


def advance_ints(t, delta=1):
    lib().call("mp_advance_ints", _p(t), t.numel(), int(delta), _stream())


def rmsnorm(x, w, eps, out=None):
    _chk(x, torch.bfloat16, "rmsnorm.x"); _chk(w, torch.float32, "rmsnorm.w")
    x2 = x.reshape(-1, x.shape[-1]); assert x2.stride(1) == 1
    if out is None:
        out = torch.empty_like(x2)
    lib().call("mp_rmsnorm_bf16", _p(x2), x2.stride(0), _p(w), _p(out), out.stride(0), x2.shape[0], x2.shape[1], float(eps),
               _stream())
    return out.view(x.shape)


def layernorm(x, w, b, eps, out=None):
    _chk(x, torch.bfloat16, "layernorm.x"); _chk(w, torch.float32, "layernorm.w")
    x2 = x.reshape(-1, x.shape[-1]); assert x2.stride(1) == 1
    if out is None:
        out = torch.empty_like(x2)
    lib().call("mp_layernorm_bf16", _p(x2), x2.stride(0), _p(w), _p(b), _p(out), out.stride(0), x2.shape[0], x2.shape[1],
               float(eps), _stream())
    return out.view(x.shape)


def rope_qk_(qkv, cos_t, sin_t, seq, heads, head_dim, pos_offset=0):
    """in place on a fused [tokens, 3*heads*head_dim] bf16 buffer; cos/sin fp32 [>=seq+pos_offset, head_dim/2]; token t of a
    sequence gets position (t % seq) + pos_offset (pos_offset = cached length when decoding)."""
    _chk(qkv, torch.bfloat16, "rope.qkv"); _chk(cos_t, torch.float32, "rope.cos")
    assert qkv.dim() == 2 and qkv.stride(1) == 1 and cos_t.is_contiguous() and sin_t.is_contiguous()
    assert cos_t.shape[0] >= seq + pos_offset and cos_t.shape[1] == head_dim // 2
    lib().call("mp_rope_qk_bf16", _p(qkv), qkv.stride(0), _p(cos_t), _p(sin_t), qkv.shape[0], seq, heads, head_dim, int(pos_offset),
               _stream())
    return qkv


def rope_interleave_qkv(qkv_w, heads, head_dim):
    """Fused qkv weight [3*H*D, K] -> the row order mp_gemm_qkv_rope_bf16 expects: inside every q and k head (D = 128 rows) the rows
    are [dims 0..31 | 64..95 | 32..63 | 96..127] (blocks of 32 of the low half followed by the matching block of the high half, the
    SwiGLU interleave applied per head); the v third is unchanged.  Pure data movement."""
    d = heads * head_dim
    assert head_dim == 128 and qkv_w.shape[0] == 3 * d
    K = qkv_w.shape[1]
    out = qkv_w.clone()
    qk = qkv_w[:2 * d].view(2 * heads, 2, 2, 32, K)            # [head, half (lo/hi), block, 32, K]
    out[:2 * d] = qk.permute(0, 2, 1, 3, 4).reshape(2 * d, K)   # [head, block, half, 32, K]
    return out


def padded_rows(rows, cols, device, dtype=torch.bfloat16):
    """A [rows, cols] view of a buffer whose row stride avoids multiples of 8 KiB: with such strides (the fused qkv output: 3 * 4096 * 2 B
    = 24 KiB) every row starts at the same offset modulo the memory channels' interleave, and the kernels that walk rows lose 10-14 %
    (scripts/gemm_pad_ab.py: K = 12288 GEMM 426 -> 375 us; scripts/attn_pad_ab.py: attention forward 77 -> 68 us).  320 elements of
    padding per row; the padding is never read or written."""
    esz = torch.empty(0, dtype=dtype).element_size()
    pad = 320 if (cols * esz) % 8192 == 0 else 0
    return torch.empty((rows, cols + pad), dtype=dtype, device=device)[:, :cols]


def gemm_fold_ok(M, N, K):
    """Whether the folded-norm GEMM entry points (row_scale= / a_row_scale=) take a call of this size (mp_gemm_fold_ok)."""
    return bool(lib().raw("mp_gemm_fold_ok")(int(M), int(N), int(K)))


def gemm_qkv_rope(a, w_interleaved, cos_t, sin_t, seq, heads, head_dim, pos_offset=0, out=None, row_scale=None):
    """qkv = a @ W^T with RoPE applied to the q and k thirds in the GEMM epilogue; `w_interleaved` = rope_interleave_qkv(W).  The result
    (standard [tokens, 3*H*D] layout) is bit-identical with gemm(a, W) followed by rope_qk_."""
    _chk(a, torch.bfloat16, "gemm_qkv_rope.a"); _chk(w_interleaved, torch.bfloat16, "gemm_qkv_rope.w"); _chk(cos_t, torch.float32, "gemm_qkv_rope.cos")
    M, K = a.shape
    N = w_interleaved.shape[0]
    assert a.stride(1) == 1 and w_interleaved.stride(1) == 1 and cos_t.is_contiguous() and sin_t.is_contiguous()
    assert cos_t.shape[0] >= seq + pos_offset and cos_t.shape[1] == head_dim // 2 and N == 3 * heads * head_dim
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    _ensure_gemm_workspace(a.device)
    t0 = GEMM_TIMER.begin() if GEMM_TIMER is not None else None
    if row_scale is not None:       # folded input norm: a = the raw residual stream, w carries the norm weight, row_scale = rstd [M] fp32
        _chk(row_scale, torch.float32, "gemm_qkv_rope.row_scale"); assert row_scale.is_contiguous() and row_scale.numel() == M
        lib().call("mp_gemm_qkv_rope_scaled_bf16", _p(a), a.stride(0), _p(w_interleaved), w_interleaved.stride(0), _p(out), out.stride(0), _p(cos_t),
                   _p(sin_t), _p(row_scale), M, N, K, int(seq), int(pos_offset), int(head_dim), _stream())
    else:
        lib().call("mp_gemm_qkv_rope_bf16", _p(a), a.stride(0), _p(w_interleaved), w_interleaved.stride(0), _p(out), out.stride(0), _p(cos_t),
                   _p(sin_t), M, N, K, int(seq), int(pos_offset), int(head_dim), _stream())
    if GEMM_TIMER is not None:
        GEMM_TIMER.end(2.0 * M * N * K, t0)
    return out


def argmax_rows(x):
    _chk(x, torch.float32, "argmax_rows.x"); assert x.dim() == 2 and x.stride(1) == 1
    out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    lib().call("mp_argmax_rows_f32", _p(x), x.stride(0), x.shape[0], x.shape[1], _p(out), _stream())
    return out


def swiglu_interleave(gate, up):
    """Pack gate/up projection weights [ff, d] into the row order the SWIGLU_PAIR epilogue expects: blocks of 32 gate rows
    followed by the 32 matching up rows.  (Pure data movement.)"""
    ff, d = gate.shape
    assert ff % 32 == 0 and up.shape == gate.shape
    return torch.stack([gate.view(ff // 32, 32, d), up.view(ff // 32, 32, d)], 1).reshape(2 * ff, d)


def swiglu_deinterleave(gu):
    two_ff, d = gu.shape
    v = gu.view(two_ff // 64, 2, 32, d)
    return v[:, 0].reshape(two_ff // 2, d), v[:, 1].reshape(two_ff // 2, d)


def swiglu(gu, out=None):
    _chk(gu, torch.bfloat16, "swiglu.gu")
    assert gu.dim() == 2 and gu.stride(1) == 1
    ff = gu.shape[1] // 2
    if out is None:
        out = torch.empty((gu.shape[0], ff), dtype=torch.bfloat16, device=gu.device)
    lib().call("mp_swiglu_bf16", _p(gu), gu.stride(0), _p(out), out.stride(0), gu.shape[0], ff, _stream())
    return out


def cast_to_bf16(x):
    _chk(x, torch.float32, "cast_to_bf16"); x = x.contiguous()
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    lib().call("mp_cast_f32_to_bf16", _p(x), _p(y), x.numel(), _stream())
    return y


def cast_to_f32(x):
    _chk(x, torch.bfloat16, "cast_to_f32"); x = x.contiguous()
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    lib().call("mp_cast_bf16_to_f32", _p(x), _p(y), x.numel(), _stream())
    return y


def add_rows(x, addend, out=None):
    """x [rows, dim] + addend [period, dim] broadcast with row % period."""
    _chk(x, torch.bfloat16, "add_rows.x"); _chk(addend, torch.bfloat16, "add_rows.addend")
    x2 = x.reshape(-1, x.shape[-1]); a2 = addend.reshape(-1, addend.shape[-1])
    assert x2.is_contiguous() and a2.is_contiguous()
    if out is None:
        out = torch.empty_like(x2)
    lib().call("mp_add_rows_bf16", _p(x2), _p(a2), _p(out), x2.shape[0], x2.shape[1], a2.shape[0], _stream())
    return out.view(x.shape)


def add3(a, b, c=None, out=None):
    _chk(a, torch.bfloat16, "add3.a")
    assert a.is_contiguous() and b.is_contiguous() and (c is None or c.is_contiguous())
    if out is None:
        out = torch.empty_like(a)
    lib().call("mp_add3_bf16", _p(a), _p(b), _p(c), _p(out), a.numel(), _stream())
    return out


# ------------------------------------------------------------------ fp32 tail -------------------------------------------------
def sgemm(a, b, trans_a=False, trans_b=False, bias=None, act=SACT_NONE, alpha=1.0, beta=0.0, out=None, split_k=1):
    """2-D or batched (3-D / 4-D leading batch dims via strides) fp32 GEMM.  Operands may be strided views as long as
    the innermost stride is 1."""
    _chk(a, torch.float32, "sgemm.a"); _chk(b, torch.float32, "sgemm.b")
    nb = a.dim() - 2
    assert b.dim() == a.dim() and nb in (0, 1, 2) and a.stride(-1) == 1 and b.stride(-1) == 1
    am, ak = (a.shape[-1], a.shape[-2]) if trans_a else (a.shape[-2], a.shape[-1])
    bk, bn = (b.shape[-1], b.shape[-2]) if trans_b else (b.shape[-2], b.shape[-1])
    assert ak == bk, f"sgemm: inner dims differ {ak} vs {bk}"
    batch = tuple(a.shape[:nb])
    if (nb == 0 and out is None and beta == 0.0 and split_k == 1 and ak >= 512 and act != SACT_SIGMOID
            and -(-am // 64) * -(-bn // 64) <= 64):
        # skinny problem with a long K: deterministic split-K — the K chunks run as a batched GEMM into [splits, M, N]
        # partials (one launch, `splits` x more workgroups), then a fixed-order column sum (+ bias, activation)
        splits = next((sp for sp in (16, 8, 4, 2) if ak % sp == 0 and (ak // sp) % 16 == 0 and ak // sp >= 128), 1)
        if splits > 1:
            chunk = ak // splits
            ws = torch.empty((splits, am, bn), dtype=torch.float32, device=a.device)
            sa = chunk * a.stride(-2) if trans_a else chunk
            sb = chunk if trans_b else chunk * b.stride(-2)
            lib().call("mp_sgemm_f32", _p(a), a.stride(-2), int(trans_a), _p(b), b.stride(-2), int(trans_b), _p(ws), bn, None,
                       am, bn, chunk, float(alpha), 0.0, SACT_NONE, splits, 1, sa, 0, sb, 0, am * bn, 0, 1, _stream())
            res = colsum_f32(ws.view(splits, am * bn)).view(am, bn)
            if bias is not None:
                res = add_f32(res, bias)
            if act != SACT_NONE:
                res = act_fwd_f32(res, act)
            return res
    if out is None:
        out = torch.empty(batch + (am, bn), dtype=torch.float32, device=a.device)
        assert beta == 0.0
    assert out.stride(-1) == 1
    nb0 = batch[0] if nb >= 1 else 1
    nb1 = batch[1] if nb == 2 else 1

    def st(t, i):
        return t.stride(i) if (i < nb and t.shape[i] > 1) else 0
    if nb == 1:
        s = [(st(t, 0), 0) for t in (a, b, out)]
    elif nb == 2:
        s = [(st(t, 0), st(t, 1)) for t in (a, b, out)]
    else:
        s = [(0, 0)] * 3
    lib().call("mp_sgemm_f32", _p(a), a.stride(-2), int(trans_a), _p(b), b.stride(-2), int(trans_b), _p(out), out.stride(-2),
               _p(bias), am, bn, ak, float(alpha), float(beta), act, nb0, nb1, s[0][0], s[0][1], s[1][0], s[1][1], s[2][0],
               s[2][1], split_k, _stream())
    return out


def layernorm_fwd_f32(x, w, b, eps):
    _chk(x, torch.float32, "layernorm_fwd_f32.x"); assert x.is_contiguous()
    rows, dim = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    lib().call("mp_layernorm_fwd_f32", _p(x), _p(w), _p(b), _p(y), _p(mean), _p(rstd), rows, dim, float(eps), _stream())
    return y, mean, rstd


def layernorm_bwd_f32(dy, x, w, mean, rstd, dw_accum, db_accum):
    assert dy.is_contiguous() and x.is_contiguous()
    rows, dim = x.numel() // x.shape[-1], x.shape[-1]
    dx = torch.empty_like(x)
    lib().call("mp_layernorm_bwd_f32", _p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(dx), _p(dw_accum), _p(db_accum), rows, dim,
               _stream())
    return dx


def softmax_fwd_f32(x, scale=1.0):
    assert x.is_contiguous()
    y = torch.empty_like(x)
    lib().call("mp_softmax_fwd_f32", _p(x), _p(y), x.numel() // x.shape[-1], x.shape[-1], float(scale), _stream())
    return y


def softmax_bwd_f32(p, dp, scale=1.0):
    assert p.is_contiguous() and dp.is_contiguous()
    dx = torch.empty_like(p)
    lib().call("mp_softmax_bwd_f32", _p(p), _p(dp), _p(dx), p.numel() // p.shape[-1], p.shape[-1], float(scale), _stream())
    return dx


def add_f32(a, b, out=None):
    """a + b where b is broadcast with period b.numel() over the flattened a."""
    _chk(a, torch.float32, "add_f32.a"); _chk(b, torch.float32, "add_f32.b")
    assert a.is_contiguous() and b.is_contiguous() and a.numel() % b.numel() == 0
    if out is None:
        out = torch.empty_like(a)
    lib().call("mp_add_f32", _p(a), _p(b), _p(out), a.numel(), b.numel(), _stream())
    return out


def act_fwd_f32(x, act):
    assert x.is_contiguous()
    y = torch.empty_like(x)
    lib().call("mp_act_fwd_f32", _p(x), _p(y), x.numel(), act, _stream())
    return y


def act_bwd_f32(dy, x, act):
    assert dy.is_contiguous() and x.is_contiguous()
    dx = torch.empty_like(x)
    lib().call("mp_act_bwd_f32", _p(dy), _p(x), _p(dx), x.numel(), act, _stream())
    return dx


def colsum_f32(x, out=None, accumulate=False):
    x2 = x.reshape(-1, x.shape[-1]); assert x2.is_contiguous()
    if out is None:
        out = torch.empty(x2.shape[1], dtype=torch.float32, device=x.device)
    lib().call("mp_colsum_f32", _p(x2), _p(out), x2.shape[0], x2.shape[1], int(accumulate), _stream())
    return out


def convt2x2_shuffle_fwd(G, bias, B, h, w, Co):
    assert G.is_contiguous()
    Y = torch.empty((B, 2 * h, 2 * w, Co), dtype=torch.float32, device=G.device)
    lib().call("mp_convt2x2_shuffle_fwd_f32", _p(G), _p(bias), _p(Y), B, h, w, Co, _stream())
    return Y


def convt2x2_shuffle_bwd(dY, B, h, w, Co):
    assert dY.is_contiguous()
    dG = torch.empty((B * h * w, Co * 4), dtype=torch.float32, device=dY.device)
    lib().call("mp_convt2x2_shuffle_bwd_f32", _p(dY), _p(dG), B, h, w, Co, _stream())
    return dG


def gather_rows_bf16_to_f32(src, idx):
    _chk(src, torch.bfloat16, "gather_rows.src"); _chk(idx, torch.int64, "gather_rows.idx")
    src2 = src.reshape(-1, src.shape[-1]); assert src2.stride(1) == 1
    out = torch.empty((idx.numel(), src2.shape[1]), dtype=torch.float32, device=src.device)
    lib().call("mp_gather_rows_bf16_to_f32", _p(src2), src2.stride(0), _p(idx), _p(out), idx.numel(), src2.shape[1], _stream())
    return out


def gather_rows_f32(src, idx):
    _chk(src, torch.float32, "gather_rows_f32.src"); _chk(idx, torch.int64, "gather_rows_f32.idx")
    assert src.is_contiguous()
    dim = src.numel() // src.shape[0]
    out = torch.empty((idx.numel(),) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    lib().call("mp_gather_rows_f32", _p(src), _p(idx), _p(out), idx.numel(), dim, _stream())
    return out


def scale_f32_(x, s):
    assert x.is_contiguous()
    lib().call("mp_scale_f32", _p(x), x.numel(), float(s), _stream())
    return x


# ------------------------------------------------------------------ mask head -------------------------------------------------
def pack_upsampler_weights(w1, w2):
    """ConvTranspose2d weights [Cin, Cout, 2, 2] -> GEMM operands of the fused upsampler: [(kh,kw,cout), cin] bf16."""
    w1p = w1.permute(2, 3, 1, 0).reshape(4 * w1.shape[1], w1.shape[0]).contiguous().to(torch.bfloat16)
    w2p = w2.permute(2, 3, 1, 0).reshape(4 * w2.shape[1], w2.shape[0]).contiguous().to(torch.bfloat16)
    return w1p, w2p


def pack_upsampler_weights_all(w1, w2):
    """-> (w1p, w2p, w1t, w2t): pack_upsampler_weights and the two transposes the backward reads, in ONE launch (mp_upsampler_pack_bf16)."""
    _chk(w1, torch.float32, "pack_upsampler.w1"); _chk(w2, torch.float32, "pack_upsampler.w2")
    assert w1.is_contiguous() and w2.is_contiguous() and w1.shape[2:] == (2, 2) and w2.shape[2:] == (2, 2)
    ci1, co1, ci2, co2 = w1.shape[0], w1.shape[1], w2.shape[0], w2.shape[1]
    dev, bf = w1.device, torch.bfloat16
    w1p, w2p = torch.empty((4 * co1, ci1), dtype=bf, device=dev), torch.empty((4 * co2, ci2), dtype=bf, device=dev)
    w1t, w2t = torch.empty((ci1, 4 * co1), dtype=bf, device=dev), torch.empty((ci2, 4 * co2), dtype=bf, device=dev)
    lib().call("mp_upsampler_pack_bf16", _p(w1), _p(w2), _p(w1p), _p(w2p), _p(w1t), _p(w2t), ci1, co1, ci2, co2, _stream())
    return w1p, w2p, w1t, w2t


def mask_upsample_fused(src, w1p, b1, ln_w, ln_b, w2p, b2, h, w, hyper=None, want_up=True, eps=1e-6):
    """src [B, h*w, 256] bf16 -> (up [B,32,4h,4w] bf16 or None, mask [B,4h,4w] f32 or None)."""
    _chk(src, torch.bfloat16, "upsample.src"); assert src.is_contiguous() and src.shape[1] == h * w and src.shape[2] == 256
    B = src.shape[0]
    up = torch.empty((B, 32, 4 * h, 4 * w), dtype=torch.bfloat16, device=src.device) if want_up else None
    mask = torch.empty((B, 4 * h, 4 * w), dtype=torch.float32, device=src.device) if hyper is not None else None
    lib().call("mp_mask_upsample_fused_bf16", _p(src), _p(w1p), _p(b1), _p(ln_w), _p(ln_b), _p(w2p), _p(b2), _p(hyper), _p(up),
               _p(mask), B, h, w, float(eps), _stream())
    return up, mask


def upsample_bwd_layout(B, T):
    """Float offsets of (dx2, dy1, a1, dy2, part) inside ONE buffer + its length: the backward's five outputs as one allocation, so that a
    consumer (the mask-tail program) addresses them all from one base."""
    sizes = (2 * B * T * 256, B * T * 256, B * T * 4 * 64, B * T * 4 * 128, (B * T // 8) * 256)
    offs, tot = [], 0
    for k in sizes:
        offs.append(tot)
        tot += -(-k // 64) * 64
    return offs, tot


def mask_upsample_fused_bwd(src, w1p, b1, ln_w, ln_b, w2p, b2, hyper, dmask, h, w, eps=1e-6, w1t=None, w2t=None, one_buffer=False):
    """Backward of mask_upsample_fused(..., hyper=...) -> (dx2 [2,B,h*w,256], dy1 [B*h*w,256], a1 [B*h*w*4,64], dy2 [B*h*w*4,128],
    part [B*h*w/8,256]); see mp_mask_upsample_fused_bwd_bf16 for what the caller finishes (two `tn` GEMMs + column sums).
    one_buffer: the five are views of one allocation at upsample_bwd_layout's offsets (returned as a sixth value)."""
    _chk(src, torch.bfloat16, "upsample_bwd.src"); _chk(dmask, torch.float32, "upsample_bwd.dmask")
    assert src.is_contiguous() and dmask.is_contiguous() and hyper.is_contiguous() and w % 16 == 0
    B, T = src.shape[0], h * w
    dev = src.device
    if w1t is None:
        w1t, w2t = w1p.t().contiguous(), w2p.t().contiguous()
    if one_buffer:
        offs, tot = upsample_bwd_layout(B, T)
        buf = torch.empty(tot, dtype=torch.float32, device=dev)
        dx2 = buf[offs[0]:offs[0] + 2 * B * T * 256].view(2, B, T, 256)
        dy1 = buf[offs[1]:offs[1] + B * T * 256].view(B * T, 256)
        a1 = buf[offs[2]:offs[2] + B * T * 256].view(B * T * 4, 64)
        dy2 = buf[offs[3]:offs[3] + B * T * 512].view(B * T * 4, 128)
        part = buf[offs[4]:offs[4] + (B * T // 8) * 256].view(B * T // 8, 256)
        lib().call("mp_mask_upsample_fused_bwd_bf16", _p(src), _p(w1p), _p(b1), _p(ln_w), _p(ln_b), _p(w2p), _p(b2), _p(hyper), _p(w1t), _p(w2t),
                   _p(dmask), _p(dx2), _p(dy1), _p(a1), _p(dy2), _p(part), B, h, w, float(eps), _stream())
        return dx2, dy1, a1, dy2, part, buf
    dx2 = torch.empty((2, B, T, 256), dtype=torch.float32, device=dev)
    dy1 = torch.empty((B * T, 256), dtype=torch.float32, device=dev)
    a1 = torch.empty((B * T * 4, 64), dtype=torch.float32, device=dev)
    dy2 = torch.empty((B * T * 4, 128), dtype=torch.float32, device=dev)
    part = torch.empty((B * T // 8, 256), dtype=torch.float32, device=dev)
    lib().call("mp_mask_upsample_fused_bwd_bf16", _p(src), _p(w1p), _p(b1), _p(ln_w), _p(ln_b), _p(w2p), _p(b2), _p(hyper), _p(w1t), _p(w2t),
               _p(dmask), _p(dx2), _p(dy1), _p(a1), _p(dy2), _p(part), B, h, w, float(eps), _stream())
    return dx2, dy1, a1, dy2, part


def py_slice_window(full, start, length):
    """Resolve masks[..., start:start+length] exactly like Python slicing does (negative starts wrap, ends clamp) —
    this is the 'crop' of postprocess_masks (model/MedPLIB.py:689-699)."""
    s, e, _ = slice(start, start + length).indices(full)
    return s, max(e - s, 0)


def postprocess_crop(in_h, in_w, input_size):
    pad_h, pad_w = in_h - input_size[0], in_w - input_size[1]
    y0, ch = py_slice_window(in_h, pad_h // 2, in_h - pad_h)
    x0, cw = py_slice_window(in_w, pad_w // 2, in_w - pad_w)
    return y0, x0, ch, cw


def bilinear_resize_fwd(x, crop, out_hw):
    """x [n, IH, IW] (f32/bf16) -> [n, OH, OW] f32."""
    assert x.is_contiguous() and x.dim() == 3
    n, IH, IW = x.shape
    y0, x0, ch, cw = crop
    out = torch.empty((n, out_hw[0], out_hw[1]), dtype=torch.float32, device=x.device)
    lib().call("mp_bilinear_resize_fwd", _p(x), _dt(x.dtype), _p(out), n, IH, IW, y0, x0, ch, cw, out_hw[0], out_hw[1], _stream())
    return out


def bilinear_resize_bwd(dout, in_hw, crop):
    assert dout.is_contiguous()
    n, OH, OW = dout.shape
    y0, x0, ch, cw = crop
    din = torch.zeros((n, in_hw[0], in_hw[1]), dtype=torch.float32, device=dout.device)
    lib().call("mp_bilinear_resize_bwd", _p(dout), _p(din), n, in_hw[0], in_hw[1], y0, x0, ch, cw, OH, OW, _stream())
    return din


def mask_losses_fwd(pred, gt, pred_iou, ce_loss, weights, offsets=None):
    """pred/gt [n, HW] f32 (or flat [sum HW_i] with int64 device `offsets` [n+1] for masks of different sizes), pred_iou [n] f32,
    ce_loss 1-elem f32 or None; weights = (ce, bce, dice, iou, focal).  Returns (out10, stats)."""
    _chk(pred, torch.float32, "mask_losses.pred"); _chk(gt, torch.float32, "mask_losses.gt")
    assert pred.is_contiguous() and gt.is_contiguous() and pred_iou.is_contiguous()
    if offsets is not None:
        _chk(offsets, torch.int64, "mask_losses.offsets")
        n, hw = offsets.numel() - 1, 0
    else:
        n, hw = pred.shape
    ws = lib().raw("mp_mask_losses_workspace")(n)
    work = torch.empty(ws, dtype=torch.uint8, device=pred.device)
    stats = torch.zeros((n, 8), dtype=torch.float32, device=pred.device)
    out = torch.empty(10, dtype=torch.float32, device=pred.device)
    lib().call("mp_mask_losses_fwd", _p(pred), _p(gt), _p(pred_iou), _p(ce_loss), n, hw, _p(offsets), *[float(w) for w in weights], _p(stats),
               _p(out), _p(work), ws, _stream())
    return out, stats


def mask_losses_bwd(pred, gt, stats, grad_scale, weights, offsets=None):
    n, hw = (offsets.numel() - 1, 0) if offsets is not None else pred.shape
    dpred = torch.empty_like(pred)
    dq = torch.empty(n, dtype=torch.float32, device=pred.device)
    lib().call("mp_mask_losses_bwd", _p(pred), _p(gt), _p(stats), _p(grad_scale), _p(dpred), _p(dq), n, hw, _p(offsets),
               *[float(w) for w in weights[1:]], _stream())
    return dpred, dq


def mask_threshold_iou(pred, gt, threshold=0.1):
    """pred [n, HW] logits (f32/bf16), gt [n, HW] f32 or None -> (uint8 mask [n,HW], int64 counts [n,4])."""
    assert pred.is_contiguous()
    n, hw = pred.shape
    bin_out = torch.empty((n, hw), dtype=torch.uint8, device=pred.device)
    counts = torch.zeros((n, 4), dtype=torch.int64, device=pred.device)
    lib().call("mp_mask_threshold_iou", _p(pred), _dt(pred.dtype), _p(gt), _p(bin_out), _p(counts), n, hw, float(threshold),
               _stream())
    return bin_out, counts


# ------------------------------------------------------------------ glue --------------------------------------------------------
SPLICE_PAD = -(2 ** 63)


def splice_rows(embed, feats, src_code, dim):
    """out[r] = embed[src_code[r]] | feats[-1-src_code[r]] | 0 (SPLICE_PAD).  src_code int64 on the GPU."""
    _chk(embed, torch.bfloat16, "splice.embed"); _chk(src_code, torch.int64, "splice.src_code")
    assert embed.is_contiguous() and (feats is None or feats.is_contiguous())
    out = torch.empty((src_code.numel(), dim), dtype=torch.bfloat16, device=embed.device)
    lib().call("mp_splice_rows_bf16", _p(embed), _p(feats), _p(src_code), _p(out), src_code.numel(), dim, _stream())
    return out


def patch_im2col(img, patch, k_padded):
    assert img.is_contiguous() and img.dim() == 4
    B, C, H, W = img.shape
    out = torch.empty((B * (H // patch) * (W // patch), k_padded), dtype=torch.bfloat16, device=img.device)
    lib().call("mp_patch_im2col", _p(img), _dt(img.dtype), _p(out), B, C, H, W, patch, k_padded, _stream())
    return out


def im2col_nhwc(x, OH, OW, stride, taps, out=None):
    """x [B,H,W,C] bf16 -> [B*OH*OW, len(taps)*C]; taps = [(dy, dx), ...]."""
    import ctypes
    _chk(x, torch.bfloat16, "im2col.x"); assert x.is_contiguous()
    B, H, W, C = x.shape
    n = len(taps)
    dy = (ctypes.c_int * n)(*[t[0] for t in taps]); dx = (ctypes.c_int * n)(*[t[1] for t in taps])
    if out is None:
        out = torch.empty((B * OH * OW, n * C), dtype=torch.bfloat16, device=x.device)
    assert out.shape == (B * OH * OW, n * C) and out.is_contiguous() and out.dtype == torch.bfloat16
    lib().call("mp_im2col_nhwc_bf16", _p(x), _p(out), B, H, W, C, OH, OW, stride, stride, n, ctypes.cast(dy, ctypes.c_void_p),
               ctypes.cast(dx, ctypes.c_void_p), _stream())
    return out


def scatter_parity(src, add, dst, B, OH, OW, C, s, py, px, DH, DW):
    lib().call("mp_scatter_parity_bf16", _p(src), _p(add), _p(dst), B, OH, OW, C, s, s, py, px, DH, DW, _stream())
    return dst


def window_partition(x, ws):
    _chk(x, torch.bfloat16, "window_partition.x"); assert x.is_contiguous()
    B, H, W, C = x.shape
    nwy, nwx = -(-H // ws), -(-W // ws)
    win = torch.empty((B * nwy * nwx, ws * ws, C), dtype=torch.bfloat16, device=x.device)
    lib().call("mp_window_partition_bf16", _p(x), _p(win), B, H, W, C, ws, _stream())
    return win


def window_unpartition_add(win, shortcut, ws):
    assert win.is_contiguous() and shortcut.is_contiguous()
    B, H, W, C = shortcut.shape
    out = torch.empty_like(shortcut)
    lib().call("mp_window_unpartition_add_bf16", _p(win), _p(shortcut), _p(out), B, H, W, C, ws, _stream())
    return out


def relpos_tables(qkv, rel_pos_h, rel_pos_w, Bw, heads, hh, ww):
    """qkv [Bw*hh*ww, 3*heads*64] bf16 -> rel_h [Bw*heads, hh*ww, hh], rel_w [Bw*heads, hh*ww, ww] fp32."""
    _chk(qkv, torch.bfloat16, "relpos.qkv"); _chk(rel_pos_h, torch.float32, "relpos.rel_pos_h")
    S = hh * ww
    rel_h = torch.empty((Bw * heads, S, hh), dtype=torch.float32, device=qkv.device)
    rel_w = torch.empty((Bw * heads, S, ww), dtype=torch.float32, device=qkv.device)
    lib().call("mp_relpos_tables_bf16", _p(qkv), qkv.stride(0), _p(rel_pos_h), _p(rel_pos_w), _p(rel_h), _p(rel_w), Bw, heads, hh, ww,
               64, _stream())
    return rel_h, rel_w


def sam_attention(qkv, qkv_bias, rel_pos_h, rel_pos_w, B, heads, grid, window, scale=None, out=None):
    """qkv [B*grid*grid, 3*heads*64] bf16 in image order -> attention output [B*grid*grid, heads*64] in image order: the four padded windows of
    side `window` (0 = global) with the decomposed rel-pos bias computed in the kernel (mp_sam_attention_bf16)."""
    _chk(qkv, torch.bfloat16, "sam_attention.qkv"); _chk(qkv_bias, torch.float32, "sam_attention.bias")
    _chk(rel_pos_h, torch.float32, "sam_attention.rel_pos_h"); _chk(rel_pos_w, torch.float32, "sam_attention.rel_pos_w")
    n = window if window else grid
    assert qkv.dim() == 2 and qkv.stride(1) == 1 and qkv.shape == (B * grid * grid, 3 * heads * 64)
    assert rel_pos_h.is_contiguous() and rel_pos_w.is_contiguous() and tuple(rel_pos_h.shape) == (2 * n - 1, 64) == tuple(rel_pos_w.shape)
    if out is None:
        out = torch.empty((B * grid * grid, heads * 64), dtype=torch.bfloat16, device=qkv.device)
    lib().call("mp_sam_attention_bf16", _p(qkv), qkv.stride(0), _p(qkv_bias), _p(rel_pos_h), _p(rel_pos_w), _p(out), out.stride(0), B, heads, grid,
               int(window), float(scale if scale is not None else 64 ** -0.5), _stream())
    return out


def sam_add_layernorm(x, w, b, eps, addend=None):
    """-> LayerNorm(x) or, with an addend [period, dim], (x + addend[row % period], LayerNorm of that sum)."""
    _chk(x, torch.bfloat16, "sam_add_layernorm.x")
    rows, dim = x.shape
    y = torch.empty_like(x)
    xs = torch.empty_like(x) if addend is not None else None
    lib().call("mp_sam_add_layernorm_bf16", _p(x), _p(addend), addend.shape[0] if addend is not None else 0, _p(xs), _p(w), _p(b), float(eps), _p(y),
               rows, dim, _stream())
    return y if addend is None else (xs, y)


def sam_layernorm_colsum(x, w, b, eps):
    """-> (LayerNorm(x) bf16, slab column sums [rows / 16, dim] fp32)."""
    _chk(x, torch.bfloat16, "sam_layernorm_colsum.x")
    rows, dim = x.shape
    xn = torch.empty_like(x)
    part = torch.empty((rows // 16, dim), dtype=torch.float32, device=x.device)
    lib().call("mp_sam_layernorm_colsum_bf16", _p(x), _p(w), _p(b), float(eps), _p(xn), _p(part), rows, dim, _stream())
    return xn, part


def sam_channel_gate(part, B, tokens, w1t, w2t):
    """slab sums [B * slabs, C] -> gate [B, C] fp32 = sigmoid(W2 relu(W1 mean)); the weights TRANSPOSED: w1t [C, hidden] = W1^T, w2t [hidden, C] = W2^T."""
    C = part.shape[1]
    assert tuple(w1t.shape) == (C, w2t.shape[0]) and w2t.shape[1] == C and w1t.is_contiguous() and w2t.is_contiguous()
    gate = torch.empty((B, C), dtype=torch.float32, device=part.device)
    lib().call("mp_sam_channel_gate_f32", _p(part), part.shape[0] // B, tokens, _p(w1t), _p(w2t), _p(gate), B, C, w1t.shape[1], _stream())
    return gate


def sam_im2col_scaled(xn, gate, B, grid, C):
    cols = torch.empty((B * (grid // 2) ** 2, 9 * C), dtype=torch.bfloat16, device=xn.device)
    lib().call("mp_sam_im2col_scaled_bf16", _p(xn), _p(gate), _p(cols), B, grid, C, _stream())
    return cols


def sam_im2col_parity4(s1, B, half, C):
    cols4 = torch.empty((4, B * half * half, 4 * C), dtype=torch.bfloat16, device=s1.device)
    lib().call("mp_sam_im2col_parity4_bf16", _p(s1), _p(cols4), B, half, C, _stream())
    return cols4


def sam_block_tail(y4, xn, x, mlp, ad_w, ad_b, ad_eps, next_w, next_b, next_eps, B, grid):
    """-> (x_out, h_out or None): mp_sam_block_tail_bf16."""
    dim = x.shape[1]
    x_out = torch.empty_like(x)
    h_out = torch.empty_like(x) if next_w is not None else None
    lib().call("mp_sam_block_tail_bf16", _p(y4), _p(xn), _p(x), _p(mlp), _p(ad_w), _p(ad_b), float(ad_eps), _p(next_w), _p(next_b), float(next_eps),
               _p(x_out), _p(h_out), B, grid, dim, _stream())
    return x_out, h_out


def token_mean(x, B, T, C):
    out = torch.empty((B, C), dtype=torch.float32, device=x.device)
    lib().call("mp_token_mean_bf16", _p(x), _p(out), B, T, C, _stream())
    return out


def scale_channels(x, gate, B, T, C):
    y = torch.empty_like(x)
    lib().call("mp_scale_channels_bf16", _p(x), _p(gate), _p(y), B, T, C, _stream())
    return y


def clip_embed(patch, cls, pos, B, n_patches, C):
    out = torch.empty((B, n_patches + 1, C), dtype=torch.bfloat16, device=patch.device)
    lib().call("mp_clip_embed_bf16", _p(patch), _p(cls), _p(pos), _p(out), B, n_patches, C, _stream())
    return out


def adaptive_avgpool_tokens(x, len_out):
    """x [n, len_in, C] bf16 -> [n, len_out, C] (nn.AdaptiveAvgPool1d over the token axis)."""
    _chk(x, torch.bfloat16, "adaptive_avgpool_tokens.x"); assert x.is_contiguous() and x.dim() == 3
    n, len_in, C = x.shape
    out = torch.empty((n, len_out, C), dtype=torch.bfloat16, device=x.device)
    lib().call("mp_adaptive_avgpool_tokens_bf16", _p(x), _p(out), n, len_in, len_out, C, _stream())
    return out


def conv3x3s2_c1_gelu(img, w, bias):
    """img [n, H, W] (bf16 or f32) -> NHWC [n, OH, OW, CO] bf16 = gelu(Conv2d(1, CO, 3, stride 2, pad 1)); w [CO, 9] f32."""
    assert img.is_contiguous() and img.dim() == 3 and img.is_cuda
    _chk(w, torch.float32, "conv3x3s2_c1.w"); _chk(bias, torch.float32, "conv3x3s2_c1.bias")
    n, H, W = img.shape
    CO = w.shape[0]
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    out = torch.empty((n, OH, OW, CO), dtype=torch.bfloat16, device=img.device)
    lib().call("mp_conv3x3s2_c1_gelu_bf16", _p(img), _dt(img.dtype), _p(w), _p(bias), _p(out), n, H, W, CO, _stream())
    return out


def region_point_mean(fmap, xy, offsets, map_index, h, w):
    """fmap [n_maps, h*w, C] bf16, xy [n_pts, 2] f32 (x, y in [0,1]), offsets int64 [n_masks+1], map_index int32 [n_masks]
    -> [n_masks, C] bf16 (extract_region_feature)."""
    _chk(fmap, torch.bfloat16, "region_point_mean.fmap"); _chk(xy, torch.float32, "region_point_mean.xy")
    _chk(offsets, torch.int64, "region_point_mean.offsets"); _chk(map_index, torch.int32, "region_point_mean.map_index")
    assert fmap.is_contiguous() and xy.is_contiguous()
    n, C = map_index.numel(), fmap.shape[-1]
    out = torch.empty((n, C), dtype=torch.bfloat16, device=fmap.device)
    lib().call("mp_region_point_mean_bf16", _p(fmap), _p(xy), _p(offsets), _p(map_index), _p(out), n, h, w, C, _stream())
    return out


def copy_rows(src, rows, dim, rows_per_batch, src_batch_rows, src_row0):
    dst = torch.empty((rows, dim), dtype=torch.bfloat16, device=src.device)
    lib().call("mp_copy_rows_bf16", _p(src), _p(dst), rows, dim, rows_per_batch, src_batch_rows, src_row0, _stream())
    return dst


# ------------------------------------------------------------------ CE / MoE ---------------------------------------------------
def cross_entropy_rows(logits, labels):
    _chk(logits, torch.float32, "ce.logits"); _chk(labels, torch.int64, "ce.labels")
    assert logits.stride(1) == 1
    out = torch.empty(logits.shape[0], dtype=torch.float32, device=logits.device)
    lib().call("mp_cross_entropy_rows_f32", _p(logits), logits.stride(0), _p(labels), logits.shape[0], logits.shape[1], _p(out), _stream())
    return out


def mean_plus(x, scale=1.0, add=None, add_scale=0.0):
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    lib().call("mp_mean_plus_f32", _p(x), x.numel(), float(scale), _p(add), 0 if add is None else add.numel(), float(add_scale),
               _p(out), _stream())
    return out


def moe_gate(x, wg):
    _chk(x, torch.bfloat16, "moe_gate.x"); _chk(wg, torch.float32, "moe_gate.wg")
    T, d = x.shape
    E = wg.shape[0]
    logits = torch.empty((T, E), dtype=torch.float32, device=x.device)
    gates = torch.empty((T, E), dtype=torch.float32, device=x.device)
    lib().call("mp_moe_gate_bf16", _p(x), x.stride(0), _p(wg), _p(logits), _p(gates), T, d, E, _stream())
    return logits, gates


def rmsnorm_gate(x, ln_w, eps, wg=None):
    """h = rmsnorm(x) * ln_w and, with wg [E, d] fp32, the MoE gate (logits, gates fp32 [T, E]) of that h in the same pass over the rows:
    bit-identical with rmsnorm() followed by moe_gate().  dim 2048 / 4096 / 8192 (callers fall back to the two kernels otherwise).
    -> (h, logits, gates) (logits / gates None without wg)."""
    _chk(x, torch.bfloat16, "rmsnorm_gate.x"); _chk(ln_w, torch.float32, "rmsnorm_gate.w")
    T, d = x.shape
    assert x.stride(1) == 1
    h = torch.empty((T, d), dtype=torch.bfloat16, device=x.device)
    E = 0 if wg is None else wg.shape[0]
    logits = gates = None
    if E:
        _chk(wg, torch.float32, "rmsnorm_gate.wg"); assert wg.is_contiguous() and wg.shape[1] == d
        logits = torch.empty((T, E), dtype=torch.float32, device=x.device)
        gates = torch.empty((T, E), dtype=torch.float32, device=x.device)
    lib().call("mp_rmsnorm_gate_bf16", _p(x), x.stride(0), _p(ln_w), float(eps), _p(h), h.stride(0), _p(wg), _p(logits), _p(gates), T, d, E,
               _stream())
    return h, logits, gates


RMSNORM_GATE_DIMS = (2048, 4096, 8192)


def rmsnorm_gate_rstd(x, ln_w, eps, wg=None):
    """The folded-norm form of rmsnorm_gate: -> (rstd [T] fp32, logits, gates) (logits / gates None without wg); the normalised rows are not written."""
    _chk(x, torch.bfloat16, "rmsnorm_gate_rstd.x")
    T, d = x.shape
    E = 0 if wg is None else wg.shape[0]
    rstd = torch.empty(T, dtype=torch.float32, device=x.device)
    logits = torch.empty((T, E), dtype=torch.float32, device=x.device) if E else None
    gates = torch.empty((T, E), dtype=torch.float32, device=x.device) if E else None
    lib().call("mp_rmsnorm_gate_rstd_bf16", _p(x), x.stride(0), _p(ln_w), float(eps), _p(wg), _p(logits), _p(gates), _p(rstd), T, d, E, _stream())
    return rstd, logits, gates


def moe_route_top1(gates, capacity, rts_uniform=None, want_slot_token=False):
    """-> (expert, slot, weight, kept, counts, l_aux[, slot_token [E, capacity] int32: token held by each slot])."""
    T, E = gates.shape
    dev = gates.device
    slot_token = torch.empty((E, int(capacity)), dtype=torch.int32, device=dev) if want_slot_token else None
    expert = torch.empty(T, dtype=torch.int32, device=dev); slot = torch.empty(T, dtype=torch.int32, device=dev)
    weight = torch.empty(T, dtype=torch.float32, device=dev)
    kept = torch.empty(E, dtype=torch.int32, device=dev); counts = torch.empty(E, dtype=torch.int64, device=dev)
    l_aux = torch.empty(1, dtype=torch.float32, device=dev)
    lib().call("mp_moe_route_top1", _p(gates), _p(rts_uniform), T, E, int(capacity), _p(expert), _p(slot), _p(weight), _p(kept),
               _p(counts), _p(l_aux), _p(slot_token), _stream())
    if want_slot_token:
        return expert, slot, weight, kept, counts, l_aux, slot_token
    return expert, slot, weight, kept, counts, l_aux


def decode_norm_gate_route(x, ln_w, eps, wg, capacity, rts_uniform=None):
    """Decode rows (T <= 8): post-attention RMSNorm + gate + top-1 routing in one launch (bit-identical with rmsnorm + moe_gate +
    moe_route_top1).  -> (h [T, d] bf16, expert, slot, weight, kept, counts, l_aux)."""
    _chk(x, torch.bfloat16, "decode_norm_gate_route.x"); _chk(ln_w, torch.float32, "decode_norm_gate_route.ln_w")
    _chk(wg, torch.float32, "decode_norm_gate_route.wg")
    T, d = x.shape
    E = wg.shape[0]
    dev = x.device
    h = torch.empty((T, d), dtype=torch.bfloat16, device=dev)
    expert = torch.empty(T, dtype=torch.int32, device=dev); slot = torch.empty(T, dtype=torch.int32, device=dev)
    weight = torch.empty(T, dtype=torch.float32, device=dev)
    kept = torch.empty(E, dtype=torch.int32, device=dev); counts = torch.empty(E, dtype=torch.int64, device=dev)
    l_aux = torch.empty(1, dtype=torch.float32, device=dev)
    lib().call("mp_decode_norm_gate_route", _p(x), x.stride(0), _p(ln_w), float(eps), _p(wg), _p(h), h.stride(0), _p(rts_uniform), T, d, E,
               int(capacity), None, _p(expert), _p(slot), _p(weight), _p(kept), _p(counts), _p(l_aux), _stream())
    return h, expert, slot, weight, kept, counts, l_aux


def gather_rows_bf16(src, idx, out=None):
    """out[r] = src[idx[r]]: bf16 rows (src any row stride), idx int64 on the device."""
    _chk(src, torch.bfloat16, "gather_rows_bf16.src")
    n, dim = idx.numel(), src.shape[-1]
    if out is None:
        out = torch.empty((n, dim), dtype=torch.bfloat16, device=src.device)
    lib().call("mp_gather_rows_bf16", _p(src), src.stride(0), _p(idx), _p(out), out.stride(0), n, dim, 0, _stream())
    return out


def scatter_rows_bf16_(dst, idx, src):
    """dst[idx[r]] = src[r] in place (rows of idx distinct)."""
    _chk(src, torch.bfloat16, "scatter_rows_bf16.src"); _chk(dst, torch.bfloat16, "scatter_rows_bf16.dst")
    lib().call("mp_gather_rows_bf16", _p(src), src.stride(0), _p(idx), _p(dst), dst.stride(0), idx.numel(), src.shape[-1], 1, _stream())
    return dst


def moe_filter_slots(slot_token, kept, needed):
    """-> (slot_token' [E, cap], kept' [E]): the routed slots whose token is marked in `needed` (uint8 [T]), compacted in slot order."""
    E, cap = slot_token.shape
    st = torch.empty_like(slot_token)
    kp = torch.empty_like(kept)
    lib().call("mp_moe_filter_slots", _p(slot_token), _p(kept), _p(needed), _p(st), _p(kp), E, cap, _stream())
    return st, kp


def gemm_batched_rows(a, w, out, m_dev, a_rows=None, c_rows=None, c_scale=None, residual=None, act=ACT_NONE, rows_stride=0, a_row_scale=None):
    """Expert GEMMs with dispatch / combine folded in.  With a_rows: a is the shared [tokens, K] matrix and expert b reads rows
    a_rows[b*rows_stride + r]; else a is [E, M, K].  With c_rows: out is the shared [tokens, N] matrix, row c_rows[...] receives
    residual[row] + c_scale[row] * bf16(acc); else out is [E, M, N(/2 for SWIGLU_PAIR)].  w [E, N, K]; m_dev int32 [E]."""
    _chk(a, torch.bfloat16, "gemm_batched_rows.a"); _chk(w, torch.bfloat16, "gemm_batched_rows.w")
    E, N, K = w.shape
    M = int(rows_stride) if a_rows is not None else a.shape[1]          # rows per expert slab (the capacity)
    lda, sa = (a.stride(0), 0) if a_rows is not None else (a.stride(1), a.stride(0))
    ldc, sc = (out.stride(0), 0) if c_rows is not None else (out.stride(1), out.stride(0))
    _ensure_gemm_workspace(a.device)
    t0 = GEMM_TIMER.begin() if GEMM_TIMER is not None else None
    if a_row_scale is not None:     # folded post-attention norm: a = the raw residual stream, w carries the norm weight, a_row_scale = rstd [tokens]
        assert a_rows is not None and c_rows is None and residual is None and act == ACT_SWIGLU_PAIR
        _chk(a_row_scale, torch.float32, "gemm_batched_rows.a_row_scale"); assert a_row_scale.numel() == a.shape[0]
        lib().call("mp_gemm_bf16_nt_batched_rows_scaled", _p(a), lda, _p(a_rows), _p(a_row_scale), _p(w), w.stride(1), w.stride(0), _p(out), ldc, sc,
                   int(rows_stride), E, int(M), N, K, _p(m_dev), _stream())
    else:
        lib().call("mp_gemm_bf16_nt_batched_rows", _p(a), lda, sa, _p(a_rows), _p(w), w.stride(1), w.stride(0), _p(out), ldc, sc, _p(c_rows),
                   _p(c_scale), _p(residual), residual.stride(0) if residual is not None else 0, int(rows_stride), E, int(M), N, K, act,
                   _p(m_dev), _stream())
    if GEMM_TIMER is not None:
        GEMM_TIMER.end(2.0 * E * M * N * K, t0, rows_dev=m_dev, slab_rows=M, flop_per_row=2.0 * N * K)
    return out


def moe_fill_dropped(x, slot, out):
    """out[t] = x[t] where slot[t] < 0 (tokens dropped by the capacity limit keep the residual stream only)."""
    T, d = x.shape
    lib().call("mp_moe_fill_dropped_bf16", _p(x), _p(slot), _p(out), T, d, _stream())
    return out


def moe_route_top2(gates, logits, capacity, noise=None):
    """DeepSpeed top2gating; returns entry arrays of length 2T (first choices, then second choices)."""
    T, E = gates.shape
    dev = gates.device
    expert = torch.empty(2 * T, dtype=torch.int32, device=dev); slot = torch.empty(2 * T, dtype=torch.int32, device=dev)
    weight = torch.empty(2 * T, dtype=torch.float32, device=dev)
    kept = torch.empty(E, dtype=torch.int32, device=dev); counts = torch.empty(E, dtype=torch.int64, device=dev)
    l_aux = torch.empty(1, dtype=torch.float32, device=dev)
    lib().call("mp_moe_route_top2", _p(gates), _p(logits), _p(noise), T, E, int(capacity), _p(expert), _p(slot), _p(weight), _p(kept),
               _p(counts), _p(l_aux), _stream())
    return expert, slot, weight, kept, counts, l_aux


def gate_noise(n, seed, offset, gumbel, device):
    out = torch.empty(n, dtype=torch.float32, device=device)
    lib().call("mp_gate_noise_f32", _p(out), n, int(seed), int(offset), 1 if gumbel else 0, _stream())
    return out


def moe_dispatch(x, expert, slot, n_experts, capacity, buf=None, top_k=1):
    T, d = x.shape
    if buf is None:
        buf = torch.empty((n_experts, capacity, d), dtype=torch.bfloat16, device=x.device)
    assert buf.dim() == 3 and buf.shape[2] == d and buf.stride(2) == 1 and buf.stride(0) == capacity * buf.stride(1)
    lib().call("mp_moe_dispatch_bf16", _p(x), x.stride(0), _p(expert), _p(slot), _p(buf), buf.stride(1), T, d, capacity, top_k, _stream())
    return buf


def moe_combine(y, expert, slot, weight, residual, capacity, top_k=1):
    T = expert.numel() // top_k
    d = y.shape[-1]
    out = torch.empty((T, d), dtype=torch.bfloat16, device=y.device)
    lib().call("mp_moe_combine_bf16", _p(y), _p(expert), _p(slot), _p(weight), _p(residual), _p(out), T, d, capacity, top_k, _stream())
    return out


def moe_residual_mix(x, moe, mlp, coef):
    """x + (moe * c0 + mlp * c1), (c0, c1) = softmax(coef[:, :2]) per row (DeepSpeed residual MoE)."""
    T, d = x.shape
    out = torch.empty((T, d), dtype=torch.bfloat16, device=x.device)
    lib().call("mp_moe_residual_mix_bf16", _p(x), _p(moe), _p(mlp), _p(coef), coef.stride(0), _p(out), T, d, _stream())
    return out


# ------------------------------------------------------------------ optimizer --------------------------------------------------
def sumsq_accum(x, out):
    lib().call("mp_sumsq_accum_f32", _p(x), x.numel(), _p(out), _stream())
    return out


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, wd, step, max_norm=0.0, grad_sumsq=None, grad_scale=1.0):
    for t in (p, g, m, v):
        _chk(t, torch.float32, "adamw"); assert t.is_contiguous()
    lib().call("mp_adamw_step_f32", _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
               float(wd), int(step), float(max_norm), _p(grad_sumsq), float(grad_scale), _stream())
