"""Training-step plumbing for the hot path: flat parameter/gradient buffers, fused AdamW(+EMA),
data-parallel gradient averaging over RCCL, and the autoencoder generator step restated from the
reference's Lightning wrapper.

Reference behaviour restated here (pytorch_lightning is not a dependency of this path):
  * AutoencoderTrainingWrapper.training_step, generator branch — training/autoencoders.py:367-527
    (encode :398, decode :415, loss assembly :165-245, ema.update :504-505, zero_grad/backward/clip/step :507-515)
  * optimizer / scheduler factories — training/utils.py:21-101 (AdamW, InverseLR)
  * DDP gradient averaging that Lightning's `ddp` strategy performs — train.py:138, :148-164

MI355X-first choices: every trainable tensor of a model lives in ONE flat fp32 buffer (and its
gradient in another), so the optimizer is one kernel launch, the EMA rides in the same launch, and
the DDP exchange is a handful of large collectives over xGMI instead of hundreds of small ones.
"""
import math

import torch
import torch.distributed as dist

from . import _caches
from . import functional as _fn


class FlatParameters:
    """Re-homes `params` into one contiguous fp32 buffer; `.grad` of each parameter is a view into a
    second flat buffer so autograd accumulates straight into it."""

    def __init__(self, params, pad_to=1):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        total = sum(sizes)
        self.numel = total
        self.padded = ((total + pad_to - 1) // pad_to) * pad_to
        self.data = torch.zeros(self.padded, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.padded, dtype=torch.float32, device=dev)
        self._views = []
        off = 0
        for p, n in zip(self.params, sizes):
            if p.dtype != torch.float32:
                raise TypeError("FlatParameters expects fp32 master parameters")
            self.data[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.data[off:off + n].view(p.shape)
            gv = self.grad[off:off + n].view(p.shape)
            p.grad = gv
            self._views.append(gv)
            off += n

    # steal=True (round 5): zero_grad() leaves `.grad = None`, so autograd's AccumulateGrad adopts each incoming gradient tensor as it is
    # (no kernel) instead of adding it into the flat view (one 5-us `add` launch per parameter: 379 per VAE generator step, 2.0 ms;
    # ~1 000 read-modify-write passes over 4.2 GB for the DiT), and gather_grads() moves them all into the flat buffer in ONE launch
    # (csrc/elementwise.hip sat_multi_copy).  Parameters touched by several backward passes between two zero_grad() calls still
    # accumulate (the second pass adds into the adopted tensor).
    steal = True

    def zero_grad(self):
        self.grad.zero_()
        for p, gv in zip(self.params, self._views):
            p.grad = None if self.steal else gv

    def fold_grads(self, indices, ops=None):
        """The gradients of parameters `indices` that autograd left outside the flat buffer (adopted tensors: steal mode; replaced
        `.grad`s otherwise) are copied into their views and `.grad` re-pointed at the view.  A few sat_multi_copy launches on the GPU
        (160 parameters each; the table is part of the launch, so a HIP-graph capture records it too); per-parameter copies when no
        kernel library serves the device."""
        todo = [i for i in indices if self.params[i].grad is not None and self.params[i].grad.data_ptr() != self._views[i].data_ptr()]
        batched = False
        if len(todo) > 1:
            srcs = [self.params[i].grad for i in todo]
            if all(g.dtype == torch.float32 and g.is_contiguous() and g.device == self.grad.device and g.numel() <= 0x7fffffff for g in srcs):      # sat_multi_copy: < 2^31 elements per entry
                try:
                    o = _fn._ops(ops)
                except Exception:      # noqa: BLE001 — no kernel library for this device (plain CPU use of the step objects)
                    o = None
                if o is not None and (self.grad.is_cuda or o.simulator):
                    o.multi_copy(srcs, [self._views[i] for i in todo])
                    batched = True
        for i in todo:
            p, gv = self.params[i], self._views[i]
            if not batched:
                gv.copy_(p.grad)
            p.grad = gv

    def gather_grads(self, ops=None):
        """fold_grads over every parameter (what an overlapped exchange has not folded bucket by bucket already); parameters without a
        gradient this step read the zeroed view."""
        self.fold_grads(range(len(self.params)), ops)
        for p, gv in zip(self.params, self._views):
            if p.grad is None:
                p.grad = gv


class GradAllReduce:
    """Data-parallel gradient exchange over the flat gradient buffer: SUM across ranks in a few large
    buckets (torch.distributed backend 'nccl' == RCCL over xGMI on ROCm; 'gloo' in CPU tests).
    The 1/world averaging is folded into the optimizer kernel's grad_scale — no extra HBM pass.

    mode 'all_reduce'     : one all_reduce per bucket
    mode 'reduce_scatter' : reduce_scatter_tensor + all_gather_into_tensor per bucket (drives every
                            xGMI link of the 8-GPU mesh in both phases; needs bucket % world == 0).  gloo has no
                            reduce_scatter: on that backend (decided by NAME, never by catching an error) the mode runs as
                            all_reduce; a failure of the RCCL collective propagates.
    comm_dtype            : None / torch.float32 exchanges the fp32 buffer in place; torch.bfloat16 exchanges a bf16 copy of each
                            bucket (half the xGMI bytes: per-link-bound rings) and writes the summed values back into
                            the fp32 buffer — the sum over ranks is then rounded to 8 mantissa bits per addend (opt-in: the
                            reference's DDP exchanges the gradients' own dtype).

    Contract of overlap=True: exactly ONE backward pass accumulates into each parameter between two finish() calls (the hook of
    a parameter firing twice raises); with gradient accumulation or several backward() calls per step use overlap=False.  close()
    removes the hooks.

    overlap=True (default): the exchange of a bucket is launched from autograd's post-accumulate hooks the moment the last
    gradient of the bucket has landed, on a side stream fenced by an event — it runs under the backward of the layers
    in front of it (what Lightning's DDP reducer does for the reference, train.py:138).  Gradients land in the flat
    buffer from its END towards its start (backward visits the layers in reverse), so buckets are cut from the end and
    fire in that order; every bucket records its own completion event on the side stream; `finish()` (called before the
    optimizer) launches whatever did not fire (parameters without a gradient this step) and makes the compute stream wait for
    each bucket's event (`wait_bucket(i)` is the per-bucket form).  overlap=False: everything in finish().

    single_rank_exchange: with an initialised process group of ONE rank the exchange is
    normally skipped; this flag runs it anyway — hooks, side stream, events and the RCCL collectives on a 1-rank communicator —
    so the whole code path executes on a single-GPU box (tests/test_train_step.py::test_single_rank_rccl_exchange_gpu,
    `bench.py --ddp-single-rank`)."""

    def __init__(self, flat: FlatParameters, group=None, bucket_bytes=256 << 20, mode="all_reduce", overlap=True, comm_dtype=None,
                 single_rank_exchange=None, native=None, ops=None):
        self.flat = flat
        self.group = group
        self.mode = mode
        if mode not in ("all_reduce", "reduce_scatter"):
            raise ValueError(f"GradAllReduce: unknown mode {mode!r}")
        initialised = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if initialised else 1
        self.rank = dist.get_rank(group) if initialised else 0
        single_rank_exchange = bool(single_rank_exchange)
        self.active = initialised and (self.world > 1 or bool(single_rank_exchange))
        self.backend = dist.get_backend(group) if initialised else None
        if comm_dtype in (None, torch.float32):
            self.comm_dtype = None
        elif comm_dtype == torch.bfloat16:
            self.comm_dtype = torch.bfloat16
        else:
            raise ValueError("GradAllReduce: comm_dtype must be None / torch.float32 / torch.bfloat16")
        n = flat.padded
        per = max(1, bucket_bytes // 4)
        per = max(self.world, (per // self.world) * self.world)
        if mode == "reduce_scatter" and n % self.world != 0:
            raise ValueError("FlatParameters(pad_to=world_size) is required for reduce_scatter mode")
        # buckets from the END of the buffer (first to be complete during backward); bucket sizes are multiples of world
        self.buckets = []
        e = n
        while e > 0:
            s_ = max(0, e - per)
            self.buckets.append((s_, e))
            e = s_
        # the bucket at the START of the buffer completes with the last kernel of the backward pass: its exchange cannot overlap
        # anything (profiles/r05_bench_ddp_single_rank_*.json: every other bucket is ready tens of milliseconds earlier) — cut it into
        # quarters so that only a quarter-size collective is exposed behind the backward
        if len(self.buckets) > 1:
            s0, e0 = self.buckets.pop()
            # rounded UP to a multiple of world: at most four pieces, every one but the last (the one at the very start of the buffer, which
            # takes what is left) world-aligned — rounding down left a fifth, unaligned remainder bucket (36 elements, world 8: 8,8,8,8,4)
            q = max(self.world, -(-(-(-(e0 - s0) // 4)) // self.world) * self.world)
            e = e0
            while e > s0:
                s_ = max(s0, e - q)
                self.buckets.append((s_, e))
                e = s_
        # ONE exchange path by default (round 5): torch.distributed's RCCL communicator.  native=True (explicit argument only — no
        # environment switch) sends the same buckets through the C-ABI's own RCCL communicator instead (csrc/comm.hip sat_allreduce_*:
        # SURVEY.md §8b; executed on the MI355X in round 5 — tests/test_train_step.py::test_native_exchange_gpu, and
        # profiles/r05_experiments/lean_ab/ddp_{torch,native}.json: 159.80 vs 159.81 ms per step); the process group is then only the
        # side channel that hands rank 0's unique id to the other ranks.  Same collectives, same streams and events.
        native = bool(native)
        self.native, self._comm, self._ops = False, None, None
        if native and self.active:
            if self.backend == "gloo":
                raise RuntimeError("GradAllReduce(native=True) exchanges device buffers over RCCL: it needs GPUs (backend 'nccl')")
            from . import ops as _ops_mod
            self._ops = ops if ops is not None else _ops_mod.get_ops()
            uid = [self._ops.allreduce_unique_id() if self.rank == 0 else None]
            if self.world > 1:
                dist.broadcast_object_list(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            self._comm = self._ops.allreduce_init(uid[0], self.world, self.rank)
            self.native = True
        self.grad_scale = 1.0 / self.world
        self.overlap = bool(overlap) and self.active
        self._hooks = []
        self._side = None
        self._fired = [False] * len(self.buckets)
        self._done = [None] * len(self.buckets)          # per-bucket completion events (side stream)
        self.launch_log = []                             # (bucket index, launched from a hook?) of the current step — read by the tests
        self._in_hook = False
        self._pending = []                               # parameters whose final gradient is not in the flat buffer yet (hooks)
        self._fold_ops = ops                             # SatOps handle for the folds (None: the product singleton, as everywhere)
        # timing=True (bench.py --ddp-single-rank / --gpus N): the per-bucket events are created with timing enabled and kept, together
        # with the events mark_backward_start() / mark_backward_end() record, until timeline() reads them — evidence of the overlap
        # (when each bucket's exchange was enqueued and when it completed, relative to the backward pass that produced it)
        self.timing = False
        self._timeline = []
        self._bwd_ev = [None, None]
        if self.overlap:
            # parameter i covers [off, off+numel): it gates every bucket it overlaps
            self._need = [0] * len(self.buckets)
            self._of_param = []
            off = 0
            for p in flat.params:
                lo, hi = off, off + p.numel()
                mine = [bi for bi, (s_, e_) in enumerate(self.buckets) if lo < e_ and hi > s_]
                for bi in mine:
                    self._need[bi] += 1
                self._of_param.append(mine)
                off = hi
            self._left = list(self._need)
            for i, p in enumerate(flat.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def _make_hook(self, i):
        def hook(param):
            # the gradient is final (post-accumulate) but may live outside the flat buffer (adopted tensors: FlatParameters.steal): it is
            # folded in together with every other pending one when a bucket completes — ONE sat_multi_copy launch per bucket instead of a
            # copy launch per parameter between the backward's kernels (each such launch costs the step 10-18 us)
            self._pending.append(i)
            complete = []
            for bi in self._of_param[i]:
                self._left[bi] -= 1
                if self._left[bi] < 0:
                    raise RuntimeError("GradAllReduce: a parameter's gradient was accumulated twice before finish() — the overlapped "
                                       "exchange supports ONE backward pass per step (use overlap=False for gradient accumulation "
                                       "or several backward() calls into the same parameters)")
                if self._left[bi] == 0:
                    complete.append(bi)
            if complete:
                self._flush_pending()
                self._in_hook = True
                try:
                    for bi in complete:
                        self._launch(bi)
                finally:
                    self._in_hook = False
        return hook

    def _flush_pending(self):
        if self._pending:
            idx, self._pending = self._pending, []
            self.flat.fold_grads(idx, self._fold_ops)

    def _exchange(self, chunk):
        buf = chunk if self.comm_dtype is None else chunk.to(self.comm_dtype)
        if self.native:
            self._ops.allreduce_bucket(self._comm, buf, mode=1 if (self.mode == "reduce_scatter" and buf.numel() % self.world == 0) else 0)
            if buf is not chunk:
                chunk.copy_(buf)
            return
        if self.mode == "reduce_scatter" and self.backend != "gloo" and buf.numel() % self.world == 0:
            k = buf.numel() // self.world
            shard = buf[self.rank * k:(self.rank + 1) * k]             # in place: output = this rank's slice of the input
            dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(buf, shard, group=self.group)       # in place: rank r's input is slice r of the output
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        if buf is not chunk:
            chunk.copy_(buf)

    def _launch(self, bi):
        if self._fired[bi]:
            return
        self._fired[bi] = True
        self.launch_log.append((bi, self._in_hook))
        s_, e_ = self.buckets[bi]
        chunk = self.flat.grad[s_:e_]
        if chunk.is_cuda and self.overlap:
            if self._side is None:
                self._side = torch.cuda.Stream(device=chunk.device)
            # timing events cannot be recorded inside a stream capture: a captured step is exchanged without a timeline
            timing = self.timing and not torch.cuda.is_current_stream_capturing()
            ev = torch.cuda.Event(enable_timing=timing)
            ev.record(torch.cuda.current_stream(chunk.device))      # the bucket's gradients are complete at this point
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                self._exchange(chunk)
                done = torch.cuda.Event(enable_timing=timing)
                done.record(self._side)
            self._done[bi] = done
            if timing:
                self._timeline.append((bi, self._in_hook, ev, done, (e_ - s_) * 4))
        else:
            self._exchange(chunk)

    def wait_bucket(self, bi):
        """The compute stream waits for bucket `bi`'s exchange only (a consumer that walks the flat buffer bucket by bucket can
        start on the first-finished ones)."""
        ev = self._done[bi]
        if ev is not None:
            torch.cuda.current_stream(self.flat.grad.device).wait_event(ev)

    def finish(self):
        """All buckets exchanged and visible to the compute stream; re-arms the hooks' counters for the next step."""
        if not self.active:
            return
        self._flush_pending()
        for bi in range(len(self.buckets)):
            self._launch(bi)
        for bi in range(len(self.buckets)):
            self.wait_bucket(bi)
        self.last_launch_log, self.launch_log = self.launch_log, []
        self._fired = [False] * len(self.buckets)
        self._done = [None] * len(self.buckets)
        if self.overlap:
            self._left = list(self._need)

    def __call__(self):
        self.finish()

    def mark_backward_start(self):
        if self.timing and self.flat.grad.is_cuda:
            self._timeline = []
            self._bwd_ev[0] = torch.cuda.Event(enable_timing=True)
            self._bwd_ev[0].record()

    def mark_backward_end(self):
        if self.timing and self.flat.grad.is_cuda:
            self._bwd_ev[1] = torch.cuda.Event(enable_timing=True)
            self._bwd_ev[1].record()

    def timeline(self):
        """(timing=True) The last step's exchange against its backward pass, in milliseconds from the backward's first kernel:
        {"backward_ms": ..., "buckets": [{"bucket", "bytes", "from_hook", "ready_ms", "done_ms"}, ...]} — `ready` = the bucket's
        gradients complete on the compute stream (its collective is enqueued behind that event on the side stream), `done` = the
        collective finished.  Synchronises the device."""
        if not (self.timing and self._bwd_ev[0] is not None and self._bwd_ev[1] is not None):
            return None
        torch.cuda.synchronize()
        t0 = self._bwd_ev[0]
        rows = [{"bucket": bi, "bytes": nbytes, "from_hook": bool(h), "ready_ms": round(t0.elapsed_time(ev), 3), "done_ms": round(t0.elapsed_time(done), 3)}
                for bi, h, ev, done, nbytes in self._timeline]
        return {"backward_ms": round(t0.elapsed_time(self._bwd_ev[1]), 3), "buckets": rows}

    def rearm(self):
        """Forget what the hooks of an ABORTED backward already did (a failed HIP-graph capture, an exception inside the step): the next
        backward starts from fresh counters.  Collectives already enqueued on the side stream are joined first — best effort: an event
        recorded inside an invalidated capture cannot be waited for, and that failure must neither replace the error that brought us
        here nor leave the counters half reset."""
        for bi in range(len(self.buckets)):
            try:
                self.wait_bucket(bi)
            except RuntimeError:
                pass
        self.launch_log = []
        self._pending = []
        self._fired = [False] * len(self.buckets)
        self._done = [None] * len(self.buckets)
        if self.overlap:
            self._left = list(self._need)

    def close(self):
        """Remove the autograd hooks (a second step object on the same parameters must not inherit live hooks) and, in native mode,
        destroy the communicator."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.overlap = False
        if self._comm is not None:
            self._ops.allreduce_finalize(self._comm)
            self._comm, self.native = None, False


def inverse_lr(step, base_lr, inv_gamma=1.0, power=1.0, warmup=0.0, final_lr=0.0):
    """InverseLR closed form (training/utils.py:52-56); `step` = scheduler.last_epoch."""
    w = 1 - warmup ** (step + 1)
    mult = (1 + step / inv_gamma) ** -power
    return w * max(final_lr, base_lr * mult)


def ema_decay(step, beta=0.9999, power=0.75, inv_gamma=1.0, update_after_step=1, min_value=0.0):
    """ema_pytorch.EMA.get_current_decay as configured by the wrapper (training/autoencoders.py:262-270)."""
    epoch = max(step - update_after_step - 1, 0)
    if epoch <= 0:
        return 0.0
    value = 1 - (1 + epoch / inv_gamma) ** -power
    return min(max(value, min_value), beta)


def clip_flat_grads(flat, max_norm, grad_scale=1.0, ops=None):
    """torch.nn.utils.clip_grad_norm_(params, max_norm) (training/autoencoders.py:491-492, :509-510) on a flat gradient buffer, without
    a host sync: the buffer holds the SUM over ranks and grad_scale = 1 / world turns its norm into that of the mean gradient the
    reference clips; the coefficient min(1, max_norm / (norm + 1e-6)) is multiplied in on the device.  Returns the (mean-gradient) norm."""
    # not torch.linalg.vector_norm: ops.sum_all says why.  The squares are formed slice by slice so the temporary stays at 256 MB whatever
    # the size of the flat buffer (4.2 GB of DiT gradients would otherwise cost a 4.2 GB scratch tensor and a full extra write pass).
    g, total, step = flat.grad, None, 1 << 26
    for lo in range(0, g.numel(), step):
        c = g[lo:lo + step]
        part = _fn.sum_all(c * c, ops)
        total = part if total is None else total + part
    norm = torch.sqrt(total) * grad_scale
    flat.grad.mul_(torch.clamp(max_norm / (norm + 1e-6), max=1.0))
    return norm


class FusedAdamW:
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction) over FlatParameters in
    one HIP launch (csrc/elementwise.hip sat_adamw_step)."""

    def __init__(self, flat: FlatParameters, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, ops=None, use_ema=False):
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, tuple(betas), eps, weight_decay
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        # EMA shadow (ema_pytorch.EMA as the wrapper configures it, training/autoencoders.py:262-270):
        # updated from the pre-step parameters inside the optimizer launch.
        self.ema = flat.data.clone() if use_ema else None
        self.t = 0
        self._ops = ops
        self.hyper_dev = None        # device (lr, bc1, bc2s, grad_scale, ema_decay): set by GraphedTrainStep

    def hyper(self, lr=None, grad_scale=1.0):
        """The per-step scalars of the NEXT step (t + 1) as the kernel wants them: (lr, 1 - b1^t, sqrt(1 - b2^t), grad_scale, ema_decay)."""
        import numpy as np
        t = np.float32(self.t + 1)
        one = np.float32(1.0)
        # rounded exactly as sat_adamw_step rounds them on the host (1.0f - powf(beta, t), sqrtf(...)): a replayed step is then
        # bit-identical to the eager launch of the same step
        bc1 = one - np.power(np.float32(self.betas[0]), t)
        bc2s = np.sqrt(one - np.power(np.float32(self.betas[1]), t))
        return (float(self.lr if lr is None else lr), float(bc1), float(bc2s), float(grad_scale),
                ema_decay(self.t + 1) if self.ema is not None else 0.0)

    def step(self, lr=None, grad_scale=1.0):
        ops = _fn._ops(self._ops)
        if self.hyper_dev is not None:
            # graph mode (GraphedTrainStep): the launch is frozen in a HIP graph, the scalars come from device memory; the owner of the
            # graph writes hyper() into `hyper_dev` before every replay and counts the steps
            ops.adamw_step_dev(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.hyper_dev, self.betas[0], self.betas[1],
                               self.eps, self.weight_decay, ema=self.ema)
            return
        self.t += 1
        ops.adamw_step(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.lr if lr is None else lr,
                       self.betas[0], self.betas[1], self.eps, self.weight_decay, self.t, grad_scale,
                       ema=self.ema, ema_decay=ema_decay(self.t) if self.ema is not None else 0.0)
        _caches.bump_params(self.flat.params)   # the kernel rewrote THESE parameters in place: copies derived from them are stale


class AutoencoderTrainStep:
    """The optimisation step of the Oobleck VAE, data-parallel over the default process group — restated from
    AutoencoderTrainingWrapper.training_step (training/autoencoders.py:367-527).

    Without a discriminator (`use_discriminator=False`, or no `loss_configs.discriminator` block): every step is a generator
    step on  w_mrstft * spectral(reals, decoded) + w_kl * kl.
    With the MS-STFT discriminator (loss_configs.discriminator of type "encodec", :440-454): steps ALTERNATE as in the reference
    (:476-483, manual optimisation) — odd global steps update the discriminator on its hinge loss, even steps update the
    autoencoder on  spectral + kl + w_adv * adversarial + w_fm * feature_matching  (:165-194).  Two restatement choices that do not
    change any update: in a discriminator step the autoencoder runs under no_grad (the reference back-propagates loss_dis into the
    autoencoder too and discards those gradients at the next opt_gen.zero_grad()), and in a generator step the discriminator's
    parameters do not require grad (the reference computes and discards their gradients the same way).
    With a teacher (`teacher_model=`, or `training.teacher_model` + `teacher_model_ckpt` as training/factory.py:31-40): the distillation
    objective of :169-179 — five terms at mrstft / 4 (latent MSE + four sum-and-difference STFT losses) instead of the reconstruction
    terms; the teacher is a frozen native autoencoder (its folded / packed weights come from the derived-weight caches)."""

    def __init__(self, autoencoder, model_config, ops=None, ddp_mode="all_reduce", bucket_bytes=64 << 20, ddp_overlap=True,
                 use_discriminator=True, ddp_comm_dtype=None, ddp_single_rank=None, teacher_model=None, ddp_native=None):
        from .auraloss import AutoencoderSpectralLoss
        tr = model_config["training"]
        self.model = autoencoder
        self.ops = ops
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.flat = FlatParameters(list(autoencoder.parameters()), pad_to=max(world, 1))
        oc = tr["optimizer_configs"]["autoencoder"]
        self.opt, self.base_lr, self.sched = self._make_opt(self.flat, oc, tr, ops, use_ema=bool(tr.get("use_ema", False)))
        lc = tr["loss_configs"]
        self._reject_unsupported(tr, lc)
        self.w_kl = lc.get("bottleneck", {}).get("weights", {}).get("kl", 1e-6)      # wrapper default: training/autoencoders.py:644-647
        tw = lc.get("time", {}).get("weights", {})
        self.w_l1, self.w_l2 = float(tw.get("l1", 0.0)), float(tw.get("l2", 0.0))    # L1Loss / MSELoss on (reals, decoded): :170-190
        self.clip_grad_norm = float(tr.get("clip_grad_norm", 0.0))      # wrapper kwarg, training/autoencoders.py:48, :491-492, :509-510
        # discriminator warm-up (training/autoencoders.py:40-57, :378-379, :394-398, :440-452, :476-483)
        self.warmup_steps = int(tr.get("warmup_steps", 0))
        self.warmup_mode = tr.get("warmup_mode", "adv")
        if self.warmup_mode not in ("adv", "full"):
            raise ValueError("warmup_mode must be 'adv' or 'full'")
        self.encoder_freeze_on_warmup = bool(tr.get("encoder_freeze_on_warmup", False))
        self.warmed_up = False
        # wrapper kwargs (training/autoencoders.py:45-46, :387-388, :411-413)
        self.force_input_mono = bool(tr.get("force_input_mono", False))
        self.latent_mask_ratio = float(tr.get("latent_mask_ratio", 0.0))
        # LossModule.decay (training/losses/losses.py:9-24): a loss's weight is multiplied by `decay` every time the loss is evaluated
        # — BEFORE it is applied, so evaluation k (0-based) of the generator losses sees weight * decay^(k + 1)
        self.spectral_decay = float(lc.get("spectral", {}).get("decay", 1.0))
        self.time_decay = float(lc.get("time", {}).get("decay", 1.0))
        sample_rate = model_config["sample_rate"]
        self.spectral = AutoencoderSpectralLoss(sample_rate, weight=lc["spectral"]["weights"]["mrstft"],
                                                **lc["spectral"]["config"]).to(self.flat.data.device)
        # teacher distillation (training/factory.py:31-40: `training.teacher_model` is a model config, `teacher_model_ckpt` its weights;
        # training/autoencoders.py:169-179: with a teacher the generator loss is FIVE terms at a quarter of the mrstft weight each —
        # latent MSE, and the sum-and-difference STFT loss on four pairs — and the per-channel L / R terms are not used)
        self.teacher = teacher_model
        if self.teacher is None and tr.get("teacher_model"):
            from .autoencoders import create_autoencoder_from_config
            ckpt = tr.get("teacher_model_ckpt")
            if ckpt is None:
                raise ValueError("teacher_model_ckpt must be specified if teacher_model is specified")
            self.teacher = create_autoencoder_from_config(tr["teacher_model"])
            self.teacher.load_state_dict(torch.load(ckpt, map_location="cpu")["state_dict"])
        if self.teacher is not None:
            from .auraloss import MultiResolutionSTFTLoss, SumAndDifferenceSTFTLoss
            self.teacher = self.teacher.eval().requires_grad_(False).to(self.flat.data.device)
            sd_cls = SumAndDifferenceSTFTLoss if autoencoder.out_channels == 2 else MultiResolutionSTFTLoss      # :141-146
            self.sdstft = sd_cls(sample_rate=sample_rate, **lc["spectral"]["config"]).to(self.flat.data.device)
            self.w_distill = float(lc["spectral"]["weights"]["mrstft"]) * 0.25
        ddp_kw = dict(bucket_bytes=bucket_bytes, mode=ddp_mode, overlap=ddp_overlap, comm_dtype=ddp_comm_dtype, single_rank_exchange=ddp_single_rank,
                      native=ddp_native, ops=ops)
        self.comm = GradAllReduce(self.flat, **ddp_kw)
        self.discriminator = None
        dcfg = lc.get("discriminator")
        if use_discriminator and dcfg is not None:
            if dcfg.get("type") != "encodec":
                raise NotImplementedError("only the 'encodec' MS-STFT discriminator is on the HIP path")
            from .discriminators import EncodecDiscriminator
            dev = self.flat.data.device
            # training/autoencoders.py:83-84: EncodecDiscriminator(in_channels=self.autoencoder.out_channels, **config)
            self.discriminator = EncodecDiscriminator(in_channels=autoencoder.out_channels, **dcfg["config"]).to(dev)
            self.w_adv = float(dcfg["weights"]["adversarial"])
            self.w_fm = float(dcfg["weights"]["feature_matching"])
            self.flat_d = FlatParameters(list(self.discriminator.parameters()), pad_to=max(world, 1))
            self.opt_d, self.base_lr_d, self.sched_d = self._make_opt(self.flat_d, tr["optimizer_configs"]["discriminator"], tr, ops, use_ema=False)
            self.comm_d = GradAllReduce(self.flat_d, **ddp_kw)
        self.global_step = 0
        self.gen_steps = self.disc_steps = 0
        self.use_disc = self.discriminator is not None      # switchable (bench.py times the generator-only step and the real step)

    @staticmethod
    def _reject_unsupported(tr, lc):
        """Options of the reference wrapper (training/autoencoders.py:60-160, :367-470) that change the objective and are NOT
        restated here must not be dropped silently."""
        bad = []
        for name in ("mrmel", "hubert"):                    # training/autoencoders.py:196-225
            if float(lc.get(name, {}).get("weights", {}).get(name, 0.0)) > 0.0:
                bad.append(f"loss_configs.{name}")
        if float(lc.get("hubert", {}).get("decay", 1.0)) != 1.0:          # (spectral / time decays are restated)
            bad.append("loss_configs.hubert.decay != 1.0")
        if bad:
            raise NotImplementedError("AutoencoderTrainStep does not restate: " + ", ".join(bad))

    @staticmethod
    def _make_opt(flat, oc, tr, ops, use_ema):
        if oc["optimizer"]["type"] != "AdamW":
            raise NotImplementedError("only AdamW (the configured optimizer) has a fused HIP step")
        ocfg = dict(oc["optimizer"]["config"])
        base_lr = ocfg.pop("lr", tr.get("learning_rate", 1e-4))
        opt = FusedAdamW(flat, base_lr, betas=ocfg.pop("betas", (0.9, 0.999)), eps=ocfg.pop("eps", 1e-8),
                         weight_decay=ocfg.pop("weight_decay", 1e-2), ops=ops, use_ema=use_ema)
        sched = oc.get("scheduler")
        if sched is not None and sched["type"] != "InverseLR":
            raise NotImplementedError("only the InverseLR scheduler is restated")
        return opt, base_lr, sched

    def current_lr(self):
        if self.sched is None:
            return self.base_lr
        return inverse_lr(self.gen_steps, self.base_lr, **self.sched["config"])      # the scheduler steps with its optimizer (:513-515)

    def _trim(self, decoded, reals):
        n = min(decoded.shape[-1], reals.shape[-1])          # trim_to_shortest (training/autoencoders.py:418)
        if decoded.shape[-1] != n or reals.shape[-1] != n:
            decoded, reals = decoded[..., :n], reals[..., :n].contiguous()
        return decoded, reals

    # ---- one optimisation step = _kind() (which update is due) + the device work of that update + _after() (host counters) ----
    def _kind(self):
        """"disc" or "gen": the update the reference's manual optimisation would run at this global step (:476-483)."""
        if self.global_step >= self.warmup_steps:
            self.warmed_up = True
        if self.use_disc and self.global_step % 2 == 1 and ((self.warmup_mode == "full" and self.warmed_up) or self.warmup_mode == "adv"):
            return "disc"
        return "gen"

    def _after(self, kind):
        if kind == "disc":
            self.disc_steps += 1
        else:
            self.gen_steps += 1
        self.global_step += 1

    def _lr(self, kind):
        if kind == "disc":
            return self.base_lr_d if self.sched_d is None else inverse_lr(self.disc_steps, self.base_lr_d, **self.sched_d["config"])
        return self.current_lr()

    def _encoder_input(self, reals):
        if self.force_input_mono and reals.shape[1] > 1:          # :387-388
            return reals.mean(dim=1, keepdim=True)
        return reals

    @staticmethod
    def _encode_kw(kw):
        return {k: v for k, v in kw.items() if k not in ("latent_mask", "teacher_noise")}

    def _mask_latents(self, latents, kw):
        """latent_mask_ratio (:411-413): zero a random subset of the latents before decoding.  kw["latent_mask"] (bool, latents'
        shape) injects the mask instead of drawing it (tests)."""
        if self.latent_mask_ratio <= 0.0:
            return latents
        mask = kw.get("latent_mask")
        if mask is None:
            mask = torch.rand_like(latents) < self.latent_mask_ratio
        return torch.where(mask, torch.zeros_like(latents), latents)

    def _decays(self):
        """Factors on the spectral / time loss weights at THIS generator step (LossModule.decay_weight runs once per evaluation,
        before the weight is applied: losses.py:18-24, :102-104)."""
        k = self.gen_steps + 1
        return self.spectral_decay ** k, self.time_decay ** k

    @property
    def step_scalars_change(self):
        """True when a value baked into the launches changes from step to step (loss-weight decay): GraphedTrainStep stays eager then."""
        return self.spectral_decay != 1.0 or (self.time_decay != 1.0 and (self.w_l1 > 0.0 or self.w_l2 > 0.0))

    def _disc_body(self, reals, kw):
        """Discriminator update (:484-497): everything that runs on the device, nothing that counts steps."""
        m = self.model
        self.flat_d.zero_grad()
        with torch.no_grad():
            latents = m.encode(self._encoder_input(reals), **self._encode_kw(kw))
            latents = self._mask_latents(latents, kw)
            decoded, reals_t = self._trim(m.decode(latents), reals)
        # scale by scale: one scale's graph alive at a time (the sum of the per-scale terms is loss(): discriminators.py:42-63)
        decoded = decoded.contiguous()
        loss_dis = torch.zeros((), device=reals.device)
        for i in range(self.discriminator.discriminators.num_discriminators):
            dis_i, _, _ = self.discriminator.scale_losses(i, reals_t, decoded, need_fm=False)
            dis_i.backward()
            loss_dis = loss_dis + dis_i.detach()
        self.flat_d.gather_grads(self.ops)
        self.comm_d()
        if self.clip_grad_norm > 0.0:
            clip_flat_grads(self.flat_d, self.clip_grad_norm, self.comm_d.grad_scale, self.ops)
        self.opt_d.step(lr=self._lr("disc"), grad_scale=self.comm_d.grad_scale)
        return {"loss": loss_dis.detach(), "discriminator_loss": loss_dis.detach()}

    def _gen_body(self, reals, kw):
        """Generator update (:498-515)."""
        m = self.model
        self.flat.zero_grad()
        enc_in = self._encoder_input(reals)
        if self.warmed_up and self.encoder_freeze_on_warmup:
            with torch.no_grad():
                latents, info = m.encode(enc_in, return_info=True, **self._encode_kw(kw))
        else:
            latents, info = m.encode(enc_in, return_info=True, **self._encode_kw(kw))
        own_latents = latents                 # the latent-distillation term sees the latents BEFORE masking (:400, :411-413)
        if self.teacher is not None:
            with torch.no_grad():             # :405-408 — the teacher's bottleneck samples too (kw["teacher_noise"] injects its draw: tests)
                t_latents = self.teacher.encode(enc_in, **({"noise": kw["teacher_noise"]} if "teacher_noise" in kw else {}))
        latents = self._mask_latents(latents, kw)
        decoded, reals_t = self._trim(m.decode(latents), reals)
        sdec, tdec = self._decays()
        if self.teacher is not None:
            n = reals_t.shape[-1]
            with torch.no_grad():             # :429-437: all three under no_grad — the last two terms below carry no gradient at all
                t_decoded = self.teacher.decode(t_latents)[..., :n].contiguous()
                own_t_decoded = self.teacher.decode(latents.detach())[..., :n].contiguous()      # own latents, teacher's decoder
                t_own_decoded = m.decode(t_latents)[..., :n].contiguous()                        # teacher's latents, own decoder
            w = self.w_distill * sdec
            # AuralossLoss passes (target, input): losses.py:111 — x = the target_key tensor, y = the input_key tensor
            terms = {"latent_distill_loss": _fn.mean_all((t_latents - own_latents) ** 2, self.ops),
                     "mrstft_loss": self.sdstft(reals_t, decoded),
                     "mrstft_loss_distill": self.sdstft(t_decoded, decoded),
                     "mrstft_loss_own_latents_teacher": self.sdstft(reals_t, own_t_decoded),
                     "mrstft_loss_teacher_latents_own": self.sdstft(reals_t, t_own_decoded)}
            out = {k: (w * v).detach() for k, v in terms.items()}
            loss = w * sum(terms.values()) + self.w_kl * info["kl"]
            out["kl_loss"] = (self.w_kl * info["kl"]).detach()
        else:
            mrstft = self.spectral(reals_t, decoded)
            if sdec != 1.0:
                mrstft = mrstft * sdec
            loss = mrstft + self.w_kl * info["kl"]
            out = {"mrstft_loss": mrstft.detach(), "kl_loss": (self.w_kl * info["kl"]).detach()}
        if self.w_l1 > 0.0:
            l1 = _fn.mean_all((reals_t - decoded).abs(), self.ops)
            loss = loss + (self.w_l1 * tdec) * l1
            out["l1_time_loss"] = ((self.w_l1 * tdec) * l1).detach()
        if self.w_l2 > 0.0:
            l2 = _fn.mean_all((reals_t - decoded) ** 2, self.ops)
            loss = loss + (self.w_l2 * tdec) * l2
            out["l2_time_loss"] = ((self.w_l2 * tdec) * l2).detach()
        if self.use_disc and self.warmed_up:      # before the warm-up ends the adversarial / feature-matching terms are zero (:441-452)
            # adversarial + feature-matching terms: their gradient w.r.t. the decoded audio is collected scale by scale on a detached
            # leaf (one scale's activations alive at a time), then enters the autoencoder's single backward pass below
            leaf = decoded.detach().contiguous().requires_grad_(True)
            loss_adv = torch.zeros((), device=reals.device)
            fm = torch.zeros((), device=reals.device)
            for p in self.flat_d.params:
                p.requires_grad_(False)
            try:
                for i in range(self.discriminator.discriminators.num_discriminators):
                    _, adv_i, fm_i = self.discriminator.scale_losses(i, reals_t, leaf)
                    (self.w_adv * adv_i + self.w_fm * fm_i).backward()
                    loss_adv, fm = loss_adv + adv_i.detach(), fm + fm_i.detach()
            finally:
                for p in self.flat_d.params:
                    p.requires_grad_(True)
            self.comm.mark_backward_start()
            torch.autograd.backward([loss, decoded], [torch.ones_like(loss), leaf.grad])
            loss = loss.detach() + self.w_adv * loss_adv + self.w_fm * fm
            out.update(loss_adv=(self.w_adv * loss_adv).detach(), feature_matching=(self.w_fm * fm).detach())
        else:
            self.comm.mark_backward_start()
            loss.backward()
        self.comm.mark_backward_end()
        self.flat.gather_grads(self.ops)
        self.comm()
        if self.clip_grad_norm > 0.0:
            clip_flat_grads(self.flat, self.clip_grad_norm, self.comm.grad_scale, self.ops)
        self.opt.step(lr=self._lr("gen"), grad_scale=self.comm.grad_scale)
        out["loss"] = loss.detach()
        return out

    def __call__(self, reals, noise=None, latent_mask=None, teacher_noise=None):
        """reals: (B, C, T) on the model's device.  Returns dict of detached loss tensors (no host sync)."""
        kw = {"noise": noise} if noise is not None else {}
        if latent_mask is not None:
            kw["latent_mask"] = latent_mask
        if teacher_noise is not None:
            kw["teacher_noise"] = teacher_noise
        kind = self._kind()
        out = self._disc_body(reals, kw) if kind == "disc" else self._gen_body(reals, kw)
        self._after(kind)
        return out


class GraphedTrainStep:
    """AutoencoderTrainStep with every update replayed from ONE HIP graph (torch.cuda.CUDAGraph): a generator step of the Oobleck VAE is
    ~1250 launches, ~400 of them a few microseconds long (weight-norm folds, weight packing, SnakeBeta constants, row sums, the gradient
    accumulation adds of ~300 parameters), and between the millisecond-long conv kernels the host cannot issue those as fast as the GPU
    retires them: 6–14 ms of a 150–160 ms step are launch gaps (profiles/EXPERIMENTS.md, round 4).  Capturing forward + loss + backward +
    gradient exchange + fused AdamW(+EMA) of one update and replaying it removes them — same kernels, same arithmetic, same order.

    What changes from step to step lives in device memory the captured launches read: the batch (copied into static buffers) and the five
    optimizer scalars (learning rate, Adam bias corrections, 1 / world, EMA decay: csrc/elementwise.hip sat_adamw_step_dev).  Host-side
    bookkeeping (step counters, the cache epochs of the rewritten parameters) runs after each replay.  One graph per kind of update the
    step can be asked for: generator before / after the discriminator warm-up, discriminator.  The first `eager_steps` calls of a kind
    run eagerly (lazy workspaces, allocator warm-up); a kind whose capture fails (a host synchronisation inside) stays eager — `fallback`
    says why.  Inputs of another shape than the captured one run eagerly too.  The returned loss tensors are the graph's static buffers:
    read them before the next call.

    Platform caveat (round 4, profiles/EXPERIMENTS.md last section): on torch 2.10 / ROCm 7 a REPLAYED torch reduction of millions of
    elements to a scalar (`.mean()`, `.sum()`, `.max()`, `.norm()`: partials + semaphores) returns stale or foreign values after a few
    replays — reproduced with torch ops alone (tools/diag_graph_reduce.py).  Nothing inside the captured updates uses one any more
    (ops.sum_all / functional.mean_all: row-sum passes); code added to `_gen_body` / `_disc_body` must keep it that way."""

    def __init__(self, stepper, eager_steps=1):
        self.stepper = stepper
        self.eager_steps = int(eager_steps)
        self.graphs, self.seen, self.fallback = {}, {}, {}
        self.replays = 0
        dev = stepper.flat.data.device
        self._hyper = {"gen": torch.zeros(5, dtype=torch.float32, device=dev)}
        if stepper.discriminator is not None:
            self._hyper["disc"] = torch.zeros(5, dtype=torch.float32, device=dev)
        self._ring = [torch.zeros(5, dtype=torch.float32).pin_memory() for _ in range(8)]
        self._ring_ev = [None] * 8
        self._slot = 0

    def _opt(self, kind):
        return self.stepper.opt_d if kind == "disc" else self.stepper.opt

    def _flat(self, kind):
        return self.stepper.flat_d if kind == "disc" else self.stepper.flat

    def _comm(self, kind):
        return self.stepper.comm_d if kind == "disc" else self.stepper.comm

    def _capture(self, key, kind, reals, noise):
        s = self.stepper
        static_reals = reals.clone()
        static_noise = noise.clone() if noise is not None else None
        kw = {"noise": static_noise} if static_noise is not None else {}
        opt = self._opt(kind)
        opt.hyper_dev = self._hyper[kind]
        graph = torch.cuda.CUDAGraph()
        # Capture must not depend on host-side cache state: a derived-weight cache HIT (an eager no_grad forward at this parameter epoch
        # just before — demo, validation) would leave the fold / pack launches out of the graph and every replay would read a stale,
        # later freed buffer.  Dropping every derived copy first makes the captured update rebuild (and therefore capture) them all.
        _caches.bump_weight_epoch()
        try:
            torch.cuda.synchronize()
            with torch.cuda.graph(graph):
                out = s._disc_body(static_reals, kw) if kind == "disc" else s._gen_body(static_reals, kw)
        except BaseException:
            # entries created during the aborted capture point at graph-pool tensors that never executed: drop them, and re-arm the
            # gradient exchange whose hooks already fired inside the aborted backward (the eager retry of this step runs them again)
            _caches.bump_weight_epoch()
            for c in (s.comm, getattr(s, "comm_d", None)):
                if c is not None and hasattr(c, "rearm"):
                    c.rearm()
            raise
        finally:
            opt.hyper_dev = None
        _caches.bump_weight_epoch()      # the pool-resident copies made during capture hold garbage until the first replay
        self.graphs[key] = (graph, static_reals, static_noise, out)

    def __call__(self, reals, noise=None):
        s = self.stepper
        kind = s._kind()
        key = (kind, bool(s.warmed_up), bool(s.use_disc), tuple(reals.shape), None if noise is None else tuple(noise.shape))
        entry = self.graphs.get(key)
        if entry is None and key not in self.fallback and s.step_scalars_change:
            self.fallback[key] = "a loss weight decays from step to step (loss_configs.*.decay != 1): the launches cannot be frozen"
        if entry is None and key not in self.fallback:
            n = self.seen.get(key, 0)
            self.seen[key] = n + 1
            if n >= self.eager_steps:
                try:
                    self._capture(key, kind, reals, noise)
                    entry = self.graphs[key]
                except Exception as e:      # noqa: BLE001 — e.g. a host synchronisation inside the update: that kind stays eager
                    self.fallback[key] = repr(e)[:300]
        if entry is None:
            kw = {"noise": noise} if noise is not None else {}
            out = s._disc_body(reals, kw) if kind == "disc" else s._gen_body(reals, kw)
            s._after(kind)
            return out
        graph, static_reals, static_noise, out = entry
        if reals.data_ptr() != static_reals.data_ptr():
            static_reals.copy_(reals)
        if static_noise is not None and noise.data_ptr() != static_noise.data_ptr():
            static_noise.copy_(noise)
        opt = self._opt(kind)
        ev = self._ring_ev[self._slot]
        if ev is not None:
            ev.synchronize()                 # the copy that read this pinned slot eight steps ago is done
        host = self._ring[self._slot]
        vals = opt.hyper(lr=s._lr(kind), grad_scale=self._comm(kind).grad_scale)
        for i, v in enumerate(vals):
            host[i] = v
        self._hyper[kind].copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._ring_ev[self._slot] = ev
        self._slot = (self._slot + 1) % len(self._ring)
        graph.replay()
        opt.t += 1
        _caches.bump_params(self._flat(kind).params)      # the replayed AdamW launch rewrote these parameters in place
        s._after(kind)
        self.replays += 1
        return out


class DiTTrainStep:
    """One optimisation step of the conditioned DiT on (pre-encoded) latents, data-parallel over the default
    process group — restated from DiffusionCondTrainingWrapper.training_step
    (training/diffusion.py:332-487): t ~ Sobol-uniform (:381-383, self.rng = SobolEngine(1, scramble=True) :253) or
    logit-normal (:384-385); alphas/sigmas (:406-409); noised = x*alpha + noise*sigma (:415); v / rectified-flow
    targets (:417-420); model(noised, t, cross_attn_cond, global_embed, cfg_dropout_prob) (:440 -> DiTWrapper.forward
    models/diffusion.py:520-557); MSE (:449, losses.py:66-91); EMA (:489-491); AdamW.  The conditioner (T5 etc.) is out of
    scope: its output tensors are inputs here.  mixed precision: `autocast_dtype=torch.bfloat16` runs the projections and
    the HIP kernels in bf16 with fp32 master weights (Lightning '--precision bf16-mixed')."""

    def __init__(self, model, lr=5e-5, betas=(0.9, 0.999), weight_decay=1e-3, cfg_dropout_prob=0.1, timestep_sampler="uniform",
                 use_ema=True, autocast_dtype=None, ops=None, ddp_mode="all_reduce", bucket_bytes=256 << 20, seed=0, ddp_overlap=True,
                 ddp_comm_dtype=None, ddp_single_rank=None):
        import math
        self._math = math
        self.model = model
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.flat = FlatParameters(list(model.parameters()), pad_to=max(world, 1))
        self.opt = FusedAdamW(self.flat, lr, betas=betas, weight_decay=weight_decay, ops=ops, use_ema=use_ema)
        self.comm = GradAllReduce(self.flat, bucket_bytes=bucket_bytes, mode=ddp_mode, overlap=ddp_overlap, comm_dtype=ddp_comm_dtype,
                                  single_rank_exchange=ddp_single_rank, ops=ops)
        self.cfg_dropout_prob = cfg_dropout_prob
        self.timestep_sampler = timestep_sampler
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.rng = torch.quasirandom.SobolEngine(1, scramble=True, seed=seed + rank)   # per-rank stream, as train.py:30-33 offsets the seed
        self.autocast_dtype = autocast_dtype
        self.objective = model.diffusion_objective
        self.global_step = 0

    def draw_timesteps(self, n, device):
        if self.timestep_sampler == "uniform":
            return self.rng.draw(n)[:, 0].to(device)
        if self.timestep_sampler == "logit_normal":
            return torch.sigmoid(torch.randn(n, device=device))
        raise NotImplementedError(f"timestep_sampler {self.timestep_sampler!r} is not restated")

    def __call__(self, latents, cross_attn_cond=None, global_embed=None, t=None, noise=None):
        math = self._math
        self.flat.zero_grad()
        if t is None:
            t = self.draw_timesteps(latents.shape[0], latents.device)
        if self.objective == "v":
            alphas, sigmas = torch.cos(t * math.pi / 2), torch.sin(t * math.pi / 2)
        else:
            alphas, sigmas = 1 - t, t
        alphas, sigmas = alphas[:, None, None], sigmas[:, None, None]
        if noise is None:
            noise = torch.randn_like(latents)
        noised = latents * alphas + noise * sigmas
        targets = noise * alphas - latents * sigmas if self.objective == "v" else noise - latents
        kw = dict(cross_attn_cond=cross_attn_cond, global_embed=global_embed, cfg_dropout_prob=self.cfg_dropout_prob)
        if self.autocast_dtype is not None:
            with torch.autocast("cuda" if latents.is_cuda else "cpu", dtype=self.autocast_dtype):
                out = self.model(noised, t, **kw)
        else:
            out = self.model(noised, t, **kw)
        loss = torch.nn.functional.mse_loss(out.float(), targets.float())
        self.comm.mark_backward_start()
        loss.backward()
        self.comm.mark_backward_end()
        self.flat.gather_grads(self.opt._ops)
        self.comm()
        self.opt.step(grad_scale=self.comm.grad_scale)
        self.global_step += 1
        return {"loss": loss.detach()}
