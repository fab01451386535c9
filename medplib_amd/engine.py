"""Data-parallel training engine with the DeepSpeed-engine call surface the reference's driver uses
(train_ds_medplib.py:439-448,599-625,687-698):

    engine, optimizer, loader, scheduler = initialize(model=..., model_parameters=..., training_data=..., collate_fn=..., config=ds_config)
    out = engine(**batch); engine.backward(out["loss"]); engine.step(); engine.global_steps
    engine.save_checkpoint(dir); engine.load_checkpoint(dir) -> (path, client_state); scheduler.get_last_lr()

One process per GPU; `torch.distributed` backend "nccl" is RCCL over xGMI.  The trainable parameters (mask decoder +
text_hidden_fcs, 21.9 M at stage-III LoRA-off) live in ONE flat fp32 buffer, their gradients in one flat fp32 buffer, so
gradient averaging is a single bucketed all-reduce launched on a side stream right after backward (ZeRO-2's
reduce-scatter + all-gather collapses to this when optimizer state is replicated — 88 MB of state is not worth sharding on
288 GB HBM), and the optimizer step is two kernels (global-norm, fused AdamW+clip)."""
import math
import os

import collections.abc
import contextlib

import torch
import torch.distributed as dist

from . import ops


class WarmupDecayLR:
    """DeepSpeed WarmupDecayLR with warmup_type 'linear' (ds_config at train_ds_medplib.py:395-404)."""

    def __init__(self, total_num_steps, warmup_min_lr=0.0, warmup_max_lr=1e-3, warmup_num_steps=0, initial_lr=None, first_step_lr="optimizer", **_):
        self.total, self.min_lr, self.max_lr = max(1, int(total_num_steps)), warmup_min_lr, warmup_max_lr
        self.warmup = max(2, int(warmup_num_steps))            # DeepSpeed clamps warmup_num_steps to >= 2
        self.last_batch_iteration = -1
        # Pinned to deepspeed==0.13.1 (the reference's requirements.txt:22; its source is absent here: parity unpinned, restated):
        # that release's WarmupLR.__init__ does NOT write a learning rate into the optimizer — only step() does — so the FIRST
        # optimizer step runs at the optimizer's own configured lr (`initial_lr` = ds_config optimizer.params.lr = args.lr), the
        # second at lr(0) = warmup_min_lr, the third at lr(1), ...  ("initialise lr in the optimizer at construction", after which the
        # first step would run at warmup_min_lr, arrived in later DeepSpeed releases.)  Without `initial_lr`: warmup_min_lr.
        # Because that reading cannot be checked against the pinned source here, it is a ds_config key, not a constant (round-3 advisor):
        # scheduler.params.first_step_lr = "optimizer" (default: the reading above — what a reference run of deepspeed==0.13.1 is
        # understood to do, and what a checkpoint's Adam state would have been built with) or "warmup_min" (the first step at
        # warmup_min_lr: the behaviour of releases that initialise the optimizer's lr at construction).  One step of difference either way.
        if first_step_lr not in ("optimizer", "warmup_min"):
            raise ValueError(f"scheduler.params.first_step_lr must be 'optimizer' or 'warmup_min', got {first_step_lr!r}")
        self.first_step_lr = first_step_lr
        self._lr = self.min_lr if (initial_lr is None or first_step_lr == "warmup_min") else float(initial_lr)
        self._initial = self._lr

    def _compute(self, it):
        if it < self.warmup:
            gamma = it / self.warmup
            return self.min_lr + (self.max_lr - self.min_lr) * gamma
        return self.max_lr * max(0.0, (self.total - it) / max(1.0, self.total - self.warmup))

    def step(self):
        self.last_batch_iteration += 1
        self._lr = self._compute(self.last_batch_iteration)

    def get_last_lr(self):
        return [self._lr]

    def state_dict(self):
        return {"last_batch_iteration": self.last_batch_iteration}

    def load_state_dict(self, sd):
        self.last_batch_iteration = sd["last_batch_iteration"]
        self._lr = self._compute(self.last_batch_iteration) if self.last_batch_iteration >= 0 else self._initial


class FlatAdamW:
    """AdamW over one flat fp32 parameter buffer (kernels mp_sumsq_accum_f32 / mp_adamw_step_f32)."""

    def __init__(self, params, lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0, max_norm=1.0):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_norm
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros(1 + 256, dtype=torch.float32, device=dev)      # [0] = the norm^2, [1:] = block partials
        off = 0
        self.offsets = {}                                       # id(param) -> (offset, numel) in the flat buffers
        for p in self.params:                                   # re-home every parameter (and its grad) into the flat buffers
            k = p.numel()
            self.offsets[id(p)] = (off, k)
            self.flat_param[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + k].view(p.shape)
            p.grad = self.flat_grad[off:off + k].view(p.shape)
            off += k
        self.step_count = 0
        self.numel = n

    def zero_grad(self):
        self.flat_grad.zero_()

    def step(self, lr=None, grad_scale=1.0):
        self.step_count += 1
        self.sumsq.zero_()
        ops.sumsq_accum(self.flat_grad, self.sumsq)
        ops.adamw_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.lr if lr is None else lr,
                       self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, self.max_norm, self.sumsq, grad_scale)
        ops.PARAM_EPOCH += 1                                   # the kernel wrote the parameters through raw pointers: no tensor _version moved

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"]); self.step_count = int(sd["step"])


class Engine:
    def __init__(self, model, params, config, training_data=None, collate_fn=None):
        self.module = model
        # a peft-style wrapper (peft_compat.PeftModel) forwards calls; stream state lives on the model underneath
        model = model.get_base_model() if hasattr(model, "get_base_model") else model
        self.core = model
        self.config = config
        opt = config.get("optimizer", {}).get("params", {})
        self.optimizer = FlatAdamW(params, lr=opt.get("lr", 1e-3), betas=tuple(opt.get("betas", (0.9, 0.95))),
                                   eps=opt.get("eps", 1e-8), weight_decay=opt.get("weight_decay", 0.0),
                                   max_norm=float(config.get("gradient_clipping", 0.0)))
        sch = config.get("scheduler", {}).get("params", None)
        self.scheduler = WarmupDecayLR(**dict(sch, initial_lr=opt.get("lr", 1e-3))) if sch else None
        self.grad_accum = int(config.get("gradient_accumulation_steps", 1))
        self.micro_steps = 0
        self.global_steps = 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.comm_stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        # "comm_backend": "rccl_capi" -> gradient buckets go through this library's own RCCL entry point (mp_allreduce_bucket) instead
        # of torch.distributed's collective; torch.distributed then only bootstraps the communicator (medplib_amd/comm.py)
        self.capi_comm = None
        if config.get("comm_backend") == "rccl_capi":
            from .comm import RcclComm
            self.capi_comm = RcclComm()
        if self.world > 1:
            # every replica starts from rank 0's trainable parameters (DeepSpeed engine._broadcast_model): one broadcast of the flat
            # fp32 buffer.  The frozen trunk is loaded / seeded identically on every rank and is not sent.
            dist.broadcast(self.optimizer.flat_param, src=0)
        # "reduce_single_rank": run the gradient collective even on a one-rank group (SUM over one rank is the identity), so the
        # communication-stream ordering can be exercised on a single GPU (tests/test_gpu_model.py)
        self.reduce_single_rank = bool(config.get("reduce_single_rank", False)) and (dist.is_initialized() or self.capi_comm is not None)
        # When every trainable tensor lives in the fp32 mask tail (stage-III "LoRA off": mask_decoder + text_hidden_fcs), the model
        # may run that tail on its own stream so it overlaps the next step's frozen trunk (MedPLIBForCausalLM.tail_side_stream);
        # backward / step then follow it onto that stream.
        if torch.cuda.is_available() and hasattr(model, "tail_side_stream") and int(config.get("overlap_mask_tail", 1)):
            names = {id(p): n for n, p in model.named_parameters()}
            tail_only = all(("mask_decoder" in names.get(id(p), "") or "text_hidden_fcs" in names.get(id(p), ""))
                            for p in self.optimizer.params)
            model.tail_side_stream = bool(tail_only)
        # Backward-overlapped, bucketed gradient reduction (DeepSpeed overlap_comm / reduce_bucket_size, train_ds_medplib.py:412-419):
        # with decoder adapters training, each layer's gradients are all-reduced from INSIDE the decoder backward as soon as the
        # layer is done (LoRAState.grad_sink), on the communication stream, while the layers below are still computing; what
        # is left (the fp32 tail, lm_head / embed_tokens, front-end modules) goes in one last bucket after backward returns.
        self._pendings, self._reduced, self._layer_ranges = [], [], {}
        self._open_marks = []               # closing events of overlapped buckets, recorded where the step waits for them
        lora = getattr(getattr(model, "model", None), "lora", None)
        # (attached on one rank as well: the sink is also how the decoder backward writes adapter gradients straight into the flat buffer)
        if lora is not None and int(config.get("overlap_comm", 1)):
            self._setup_layer_buckets(lora)
        self._timing = None                                     # enable_bucket_timing(): HIP-event pairs around backward / every bucket
        # Expert-data-parallel gradient semantics (DeepSpeed: `split_params_into_different_moe_groups_for_optimizer`, train_ds_medplib.py:
        # 422-434 + engine allreduce of expert gradients over the expert-data-parallel group).  Every rank here holds all experts' adapter
        # parameters but produces gradients only for the experts it owns, so the SUM all-reduce over the world adds each expert's gradient
        # over exactly its expert-data-parallel group (world / ep ranks); DeepSpeed then divides by THAT group's size, everything else by
        # the world size.  With grad_scale = 1 / world applied to the whole buffer, expert ranges are multiplied by ep first
        # ("deepspeed", default: what a reference checkpoint's Adam state was built with); "world" keeps the gradient of the mean loss.
        self.expert_grad_mult = None
        llm = getattr(getattr(model, "model", None), "llm", None)
        ep_obj = getattr(llm, "ep", None)
        self.ep_size = int(getattr(ep_obj, "ep", 0) or config.get("ep_size", 1))
        mode = config.get("expert_grad_scaling", "deepspeed")
        if mode not in ("deepspeed", "world"):
            raise ValueError(f"expert_grad_scaling: 'deepspeed' or 'world', not {mode!r}")
        if self.ep_size > 1 and mode == "deepspeed":
            names = {id(p): n for n, p in model.named_parameters()}
            if lora is not None:
                names.update({id(p): n for n, p in zip(lora.names, lora.params)})
            mult = torch.ones(self.optimizer.numel, dtype=torch.float32)
            self.expert_param_names = []
            for p in self.optimizer.params:
                n = names.get(id(p), "")
                if ".deepspeed_experts." in n:                   # expert tensors are told apart by their (reference) names
                    off, k = self.optimizer.offsets[id(p)]
                    mult[off:off + k] = float(self.ep_size)
                    self.expert_param_names.append(n)
            if self.expert_param_names:
                self.expert_grad_mult = mult.to(self.optimizer.flat_grad.device)
        self.training_dataloader = None
        if training_data is not None:
            sampler = None
            if self.world > 1:
                sampler = torch.utils.data.distributed.DistributedSampler(training_data, num_replicas=self.world, rank=self.rank)
            self.training_dataloader = torch.utils.data.DataLoader(
                training_data, batch_size=int(config.get("train_micro_batch_size_per_gpu", 1)), shuffle=(sampler is None),
                sampler=sampler, collate_fn=collate_fn, drop_last=True)

    # DeepSpeed-engine surface -------------------------------------------------------------------------------
    def __call__(self, **batch):
        return self.module(**batch)

    def train(self, mode=True):
        self.module.train(mode); return self

    def eval(self):
        self.module.eval(); return self

    def is_gradient_accumulation_boundary(self):
        return (self.micro_steps + 1) % self.grad_accum == 0

    def backward(self, loss):
        """autograd backward (grads accumulate straight into the flat buffer; `loss` = the loss tensor as with DeepSpeed, or the model's
        output dict, whose "loss" is then taken without ordering the caller's stream behind the mask tail), then — at an accumulation boundary — one
        bucketed all-reduce of the flat gradient on the communication stream (collective C1)."""
        if isinstance(loss, collections.abc.Mapping):             # the step's output dict itself (medplib.StreamOrderedLosses: no cross-stream wait)
            loss = loss.raw("loss") if hasattr(loss, "raw") else loss["loss"]
        with self._tail_ctx():
            if self.grad_accum > 1:
                loss = loss / self.grad_accum
            ev = self._mark("backward")
            loss.backward()
            self._mark_end(ev)
            if self.is_gradient_accumulation_boundary():
                self.launch_grad_reduce()

    # bucket timing (bench.py: "all-reduce us vs tail-backward us" per step) ---------------------------------------------------
    def enable_bucket_timing(self):
        """From now on every backward pass and every gradient bucket is bracketed by a HIP-event pair on the stream it runs on."""
        self._timing = {"backward": [], "allreduce": [], "bytes": 0}

    def disable_bucket_timing(self):
        self._timing = None

    def _mark(self, kind, nbytes=0):
        if self._timing is None or not torch.cuda.is_available():
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        self._timing[kind].append((a, b))
        self._timing["bytes"] += nbytes
        return b

    @staticmethod
    def _mark_end(ev):
        if ev is not None:
            ev.record()

    def bucket_timing_summary(self, steps):
        """-> {"tail_backward_us", "allreduce_us", "buckets_per_step", "allreduce_MB_per_step"} averaged per optimizer step (events
        read after a device synchronisation); None when timing was never enabled."""
        if self._timing is None:
            return None
        torch.cuda.synchronize()
        tot = {k: sum(a.elapsed_time(b) for a, b in self._timing[k]) * 1e3 for k in ("backward", "allreduce")}
        n = max(1, steps)
        return {"tail_backward_us": round(tot["backward"] / n, 1), "allreduce_us": round(tot["allreduce"] / n, 1),
                "buckets_per_step": len(self._timing["allreduce"]) / n, "allreduce_MB_per_step": round(self._timing["bytes"] / n / 1e6, 2)}

    def _tail_ctx(self):
        """The stream the model put the trainable tail on for this step (or the caller's stream)."""
        st = getattr(self.core, "active_tail_stream", None)
        return torch.cuda.stream(st) if st is not None else contextlib.nullcontext()

    def sync_side_streams(self):
        if hasattr(self.core, "sync_side_streams") and torch.cuda.is_available():
            self.core.sync_side_streams()

    def _setup_layer_buckets(self, lora):
        """Flat-buffer ranges of every decoder layer's trainable tensors (contiguous runs merged), and the sink that fills them."""
        by_layer = {}
        for n, p in zip(lora.names, lora.params):
            if n.startswith("model.layers.") and id(p) in self.optimizer.offsets:
                by_layer.setdefault(int(n.split(".")[2]), []).append(self.optimizer.offsets[id(p)])
        for i, spans in by_layer.items():
            merged = []
            for off, k in sorted(spans):
                if merged and merged[-1][1] == off:
                    merged[-1][1] = off + k
                else:
                    merged.append([off, off + k])
            self._layer_ranges[i] = [tuple(m) for m in merged]
        self._lora = lora
        lora.grad_sink = self._sink
        # whether a layer's sink starts a collective on the layer's gradients (the decoder backward's weight-gradient side stream, when on, is
        # waited for first); on one rank without the debug reduce nothing reads them before the end of backward
        lora.sink_reduces = self.world > 1 or self.reduce_single_rank

    def _sink(self, layer, named_grads):
        """LoRAState.grad_sink: accumulate layer `layer`'s gradients into the flat buffer (they add up over the micro-steps of a
        gradient-accumulation window) and, at the boundary micro-step, start their all-reduce."""
        lo = self._lora
        for n, g in named_grads.items():
            p = lo.params[lo.index[n]]
            p.grad.add_(g.reshape(p.grad.shape))
        if (self.world > 1 or self.reduce_single_rank) and self.is_gradient_accumulation_boundary():
            for s, e in self._layer_ranges.get(layer, ()):
                self._reduce_range(s, e)

    def _reduce_range(self, s, e):
        """SUM all-reduce of flat_grad[s:e].  With layer buckets (decoder adapters training) the collective goes to the communication
        stream so it overlaps the backward of the layers below; WITHOUT them (stage III: everything trainable sits in the mask tail) the
        bucket follows the tail backward and precedes AdamW on the stream they run on — there is nothing to overlap it with, and a
        further stream is not free: measured on one GPU (one-rank RCCL group, `MP_BENCH_FORCE_DIST=1`), hopping to a fifth stream and back
        cost the DECODER 4.5 ms per step (61 -> 65.5 ms) although the collective itself took 35 us — the runtime multiplexes streams onto a
        few hardware queues, and a queue entry that waits for the (late-running) tail backward holds back whatever shares its queue."""
        buf = self.optimizer.flat_grad[s:e]
        # torch.distributed's RCCL backend runs every collective on the process group's OWN stream (ordered behind the calling stream at
        # the call, asynchronous to it afterwards): that is already the overlap the layer buckets want, and measured free (61.6 vs 61.7 ms).
        # Only the C-ABI communicator, which runs on the stream it is given, needs the hop — and only when there is something to overlap.
        own_stream = self.comm_stream is not None and self.capi_comm is not None and bool(self._layer_ranges)
        if os.environ.get("MP_ENGINE_COMM_STREAM") == "1":       # A/B: always hop to the communication stream (the pre-round-3 behaviour)
            own_stream = self.comm_stream is not None
        ctx = torch.cuda.stream(self.comm_stream) if own_stream else contextlib.nullcontext()
        if own_stream:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
        with ctx:
            ev = self._mark("allreduce", buf.numel() * 4)
            if os.environ.get("MP_ENGINE_FAKE_REDUCE") == "1":   # debug: the stream choreography without the collective
                buf.mul_(1.0)
                work = None
            elif self.capi_comm is not None:                     # stream-ordered: nothing to wait on but the stream itself
                self.capi_comm.all_reduce_(buf)
                work = None
            else:                                                # torch.distributed ("nccl" = RCCL; gloo in the CPU tests)
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
                if ev is not None and torch.cuda.is_available() and not self._layer_ranges:
                    # the collective runs on the process group's own stream: order this stream behind it (a stream wait, the host does
                    # not block) so the closing event brackets the RCCL kernel.  ONLY for the single bucket of stage III, which AdamW
                    # waits for next anyway: with layer buckets this wait would put every bucket between two layers' backward and undo
                    # the overlap inside the very region that is being timed (round-3 advisor) — those buckets close their event where
                    # the step waits for them (wait_grad_reduce), i.e. they report issue -> needed, an upper bound of the collective.
                    work.wait()
                elif ev is not None and work is not None:
                    self._open_marks.append(ev)
                    ev = None
            self._mark_end(ev)
        self._pendings.append((work, own_stream))
        self._reduced.append((s, e))

    def launch_grad_reduce(self):
        """SUM all-reduce of whatever part of the flat gradient buffer no layer bucket has covered yet — all of it (one bucket)
        when nothing in the decoder trains.  Averaged by grad_scale = 1/world inside the AdamW kernel."""
        if self.world == 1 and not self.reduce_single_rank:
            return
        pos = 0
        for s, e in sorted(self._reduced) + [(self.optimizer.numel, self.optimizer.numel)]:
            if s > pos:
                self._reduce_range(pos, s)
            pos = max(pos, e)
        self._reduced = []

    def apply_expert_grad_scaling(self):
        """After the reduction, before the clip norm and AdamW: expert ranges x ep (see __init__); one elementwise pass, only when
        ep_size > 1 and expert tensors train."""
        if self.expert_grad_mult is not None:
            self.optimizer.flat_grad.mul_(self.expert_grad_mult)

    def wait_grad_reduce(self):
        if self._pendings:
            hop = False
            for w, own_stream in self._pendings:
                if w is not None:
                    w.wait()
                hop |= own_stream
            if hop:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
            self._pendings = []
            for ev in self._open_marks:
                ev.record()
            self._open_marks = []

    def step(self):
        boundary = self.is_gradient_accumulation_boundary()
        self.micro_steps += 1
        if not boundary:
            return
        # DeepSpeed order (engine._take_model_step): optimizer.step() with the lr currently set, THEN lr_scheduler.step(): optimizer
        # step 1 runs at the optimizer's configured lr (see WarmupDecayLR.__init__), step k >= 2 at lr(k - 2)
        lr = self.scheduler.get_last_lr()[0] if self.scheduler is not None else self.optimizer.lr
        with self._tail_ctx():
            self.wait_grad_reduce()
            self.apply_expert_grad_scaling()
            self.optimizer.step(lr=lr, grad_scale=1.0 / self.world)         # SUM all-reduce -> mean
            self.optimizer.zero_grad()
        if self.scheduler is not None:
            self.scheduler.step()
        self.global_steps += 1

    def get_lr(self):
        return self.scheduler.get_last_lr() if self.scheduler else [self.optimizer.lr]

    # checkpoints: <dir>/latest holds "global_step<N>" like DeepSpeed (train_ds_medplib.py:453-470) -------------------
    def save_checkpoint(self, save_dir, client_state=None):
        tag = f"global_step{self.global_steps}"
        self.sync_side_streams()
        if self.rank == 0:
            os.makedirs(os.path.join(save_dir, tag), exist_ok=True)
            module_sd = {n: p.detach().cpu() for n, p in self.module.named_parameters() if p.requires_grad}
            torch.save({"module": module_sd, "optimizer": {k: (v.cpu() if torch.is_tensor(v) else v)
                                                           for k, v in self.optimizer.state_dict().items()},
                        "lr_scheduler": self.scheduler.state_dict() if self.scheduler else None,
                        "global_steps": self.global_steps, "micro_steps": self.micro_steps, "client_state": client_state or {}},
                       os.path.join(save_dir, tag, "mp_rank_00_model_states.pt"))
            with open(os.path.join(save_dir, "latest"), "w") as f:
                f.write(tag)
        if self.world > 1:
            dist.barrier()

    def load_checkpoint(self, load_dir):
        self.sync_side_streams()
        latest = os.path.join(load_dir, "latest")
        if not os.path.exists(latest):
            return None, None
        tag = open(latest).read().strip()
        path = os.path.join(load_dir, tag, "mp_rank_00_model_states.pt")
        ck = torch.load(path, map_location="cpu")
        named = dict(self.module.named_parameters())
        for n, v in ck["module"].items():
            named[n].data.copy_(v)
        ops.PARAM_EPOCH += 1
        self.optimizer.load_state_dict(ck["optimizer"])
        if self.scheduler and ck.get("lr_scheduler"):
            self.scheduler.load_state_dict(ck["lr_scheduler"])
        self.global_steps, self.micro_steps = ck["global_steps"], ck["micro_steps"]
        return path, ck.get("client_state", {})


def initialize(model=None, model_parameters=None, training_data=None, collate_fn=None, config=None, **_):
    """deepspeed.initialize look-alike: -> (engine, optimizer, training_dataloader, lr_scheduler)."""
    base = model.get_base_model() if hasattr(model, "get_base_model") else model
    if hasattr(base, "resolve_training_plan"):
        # the reference's module surface (medplib_amd/surface.py): the driver hands over `model.parameters()` (or MoE param groups,
        # train_ds_medplib.py:422-436) with requires_grad flags set; they become this build's training state here
        params = base.resolve_training_plan(model_parameters)
    else:
        params = list(model_parameters) if model_parameters is not None else [p for p in model.parameters() if p.requires_grad]
    eng = Engine(model, params, config or {}, training_data, collate_fn)
    return eng, eng.optimizer, eng.training_dataloader, eng.scheduler


def split_params_into_different_moe_groups_for_optimizer(param_groups, max_group_size=None):
    """`deepspeed.moe.utils.split_params_into_different_moe_groups_for_optimizer` (train_ds_medplib.py:422-431): DeepSpeed moves expert
    parameters (tagged allreduce=False) into their own optimizer groups so they reduce over the expert-data-parallel group.  Here the
    grouping is a property of the engine: with ep_size > 1 it tells expert tensors apart by the `.deepspeed_experts.` level of their
    names and gives their gradient ranges DeepSpeed's expert-data-parallel scaling (Engine.__init__: `expert_grad_mult`,
    tests/test_host_logic.py 4-rank gloo test), so the call only normalises its argument to a list of group dicts;
    `initialize()` reads the requires_grad flags off the model."""
    if isinstance(param_groups, dict):
        return [param_groups]
    return list(param_groups)


def init_distributed(dist_backend="nccl"):
    """One process per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher (torchrun / deepspeed)."""
    if dist.is_initialized() or int(os.environ.get("WORLD_SIZE", "1")) == 1:
        return
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if dist_backend == "nccl":
        torch.cuda.set_device(local)
    dist.init_process_group(backend=dist_backend)


class AverageMeterPack:
    """All scalar meters of one logging interval reduced with ONE all-reduce (the reference issues 12 — C5)."""

    def __init__(self, names, device):
        self.names = list(names)
        self.buf = torch.zeros(2 * len(self.names), dtype=torch.float64, device=device)   # [sums..., counts...]

    def update(self, name, value, n=1):
        i = self.names.index(name)
        self.buf[i] += float(value) * n
        self.buf[len(self.names) + i] += n

    def update_many(self, values, n=1):
        """values: {name: 0-dim device tensor}.  Accumulates on the device — no host synchronisation per step (the reference's
        AverageMeter.update(loss.item()) stalls the host on every micro-batch)."""
        k = len(self.names)
        idx = [self.names.index(name) for name in values]
        vals = torch.stack([values[name].detach().reshape(()).to(self.buf.device, torch.float64) for name in values])
        self.buf[idx] += vals * n
        self.buf[[k + i for i in idx]] += n

    def all_reduce(self):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM)
        k = len(self.names)
        return {n: (self.buf[i] / self.buf[k + i].clamp(min=1)).item() for i, n in enumerate(self.names)}
