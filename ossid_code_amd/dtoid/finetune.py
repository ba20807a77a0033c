"""Online finetuning of DTOID on MI355X: flat parameter/gradient buffers, a one-launch AMSGrad step, and
data-parallel gradient averaging over RCCL.

Reference behaviour being reproduced (/root/reference/python/ossid/scripts/online_learning.py):
  optimizer  torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-6, amsgrad=True)          :258-263
  step       model.train(); out = model(batch); optimizer.zero_grad(); out['loss'].backward(); optimizer.step()
                                                                                                      :650-679
  multi-GPU  none in the online loop; the contract is PyTorch-Lightning DDP's (train.py:93-102): per-rank BatchNorm
             statistics, gradients averaged over ranks every step.

MI355X-first layout: all 34 M parameters live in ONE contiguous fp32 buffer (every nn.Parameter is a view into it)
and all gradients in a second one. The optimizer is then a single HBM-bound kernel over 5 x 136 MB instead of
~600 x 5 small launches, and the DDP exchange is a handful of large all-reduces on the flat gradient buffer
(xGMI rings are per-link bound: few big messages, not one per tensor), issued asynchronously and waited once.
Parameters that never receive a gradient (the SqueezeNet classifier and 3-channel stem the reference keeps but never
runs, network.py:199-209) sit in the tail of the buffers and are left untouched, exactly as torch.optim.Adam skips
parameters whose .grad is None.
"""
import torch

from .. import _lib

_UNUSED_MARKERS = (".backbone.classifier.", ".backbone.features.0.")


def _is_unused(name):
    return any(m in "." + name for m in _UNUSED_MARKERS)


class FlatParams:
    """Re-homes a module's parameters and gradients into two flat buffers (used parameters first)."""

    def __init__(self, module, unused_filter=_is_unused):
        seen, used, unused = set(), [], []
        for name, p in module.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            (unused if unused_filter(name) else used).append((name, p))
        self.entries = used + unused
        dev, dt = self.entries[0][1].device, self.entries[0][1].dtype
        n_used = sum(p.numel() for _, p in used)
        self.n_used = (n_used + 3) // 4 * 4                      # the fused step walks float4s
        total = self.n_used + sum(p.numel() for _, p in unused)
        self.total = (total + 3) // 4 * 4
        self.param = torch.zeros(self.total, dtype=dt, device=dev)
        self.grad = torch.zeros(self.total, dtype=dt, device=dev)
        off = 0
        self.offsets = {}
        self._views = []
        for i, (name, p) in enumerate(self.entries):
            if i == len(used):
                off = self.n_used
            n = p.numel()
            view = self.param[off:off + n].view_as(p)
            view.copy_(p.data)
            p.data = view
            gview = self.grad[off:off + n].view_as(p)
            p.grad = gview
            self._views.append(gview)
            self.offsets[name] = (off, n)
            off += n
        # weight-gradient kernels of the hand-written training path write straight into these views when they can
        # (train_ops.grad_home): the per-step gather then has nothing to copy for the 34 M convolution weights
        if self.param.is_cuda:
            from . import train_ops
            for (_, p), v in zip(self.entries[:len(used)], self._views):
                train_ops.register_grad_home(p, v)

    def zero_grad(self):
        self.grad.zero_()

    # ---- "scatter" protocol: backward with p.grad = None, then ONE multi-tensor copy into the flat buffer -------------
    # With p.grad preset to a view of the flat buffer autograd ACCUMULATES (p.grad += g): one add kernel per parameter
    # per step (~600 launches, 2.6 ms of the 57 ms step). With p.grad = None it hands over the freshly computed tensor
    # at no cost, and torch._foreach_copy_ packs all of them into the flat buffer in a handful of launches.
    def detach_grads(self, collect=False):
        """collect=True: post-accumulate hooks note, WHILE backward runs, which gradients need a copy into the flat buffer
        (gather_grads then has no loop over the ~700 parameters left to do at the end of the step, where the device has
        run dry and every host microsecond in front of the optimizer launch is step time: 0.9 ms of idle device in
        profiles/r04_finetune_step_list.txt before this)."""
        for _, p in self.entries:
            p.grad = None
        if collect:
            self._install_hooks()
            self._c_dst, self._c_src, self._c_seen = [], [], set()
        self._collecting = bool(collect)

    def _install_hooks(self):
        if self.__dict__.get("_hooks_installed"):
            return
        self._hooks_installed = True
        self._collecting = False

        def make(i, v, vp):
            def hook(p):
                if self._collecting:
                    g = p.grad
                    if g is not None:
                        self._c_seen.add(i)
                        if g.data_ptr() != vp:
                            self._c_dst.append(v)
                            self._c_src.append(g)
            return hook
        for i, ((_, p), v) in enumerate(zip(self.entries, self._views)):
            p.register_post_accumulate_grad_hook(make(i, v, v.data_ptr()))
        n_used_entries = sum(1 for name, _ in self.entries if self.offsets[name][0] < self.n_used)
        self._used_idx = frozenset(range(n_used_entries))

    def gather_grads(self, reattach=True):
        """After backward with p.grad = None: every gradient into its slice of the flat buffer (multi-tensor copies; nothing
        to do for a gradient a kernel already wrote in place, train_ops.grad_home), zeros for a USED parameter that got
        none this step (rare: an unused branch), then p.grad = the flat views again. The never-used parameters in the
        buffer's tail are left alone: the optimizer does not read them.
        reattach=False: leave `p.grad = view` to a later reattach_grads() -- 700 attribute writes that nothing on the device
        waits for; finetune_step does them AFTER it has launched the optimizer (the device is idle at the end of a step:
        every host microsecond in front of that launch is step time)."""
        if self.__dict__.get("_collecting"):
            # the hooks did the walk during backward; a USED parameter no hook fired for got no gradient this step
            self._collecting = False
            dst, src = self._c_dst, self._c_src
            zero = [self._views[i] for i in self._used_idx - self._c_seen]
            self._c_dst = self._c_src = self._c_seen = None
            if zero:
                torch._foreach_zero_(zero)
            if dst:
                torch._foreach_copy_(dst, src)
            if reattach:
                self.reattach_grads()
            return
        dst, src, zero = [], [], []
        n_used_entries = self.__dict__.get("_n_used_entries")
        if n_used_entries is None:
            n_used_entries = self._n_used_entries = sum(1 for name, _ in self.entries if self.offsets[name][0] < self.n_used)
            self._view_ptrs = [v.data_ptr() for v in self._views]
        for i, ((_, p), v, vp) in enumerate(zip(self.entries, self._views, self._view_ptrs)):
            g = p.grad
            if g is None:
                if i < n_used_entries:
                    zero.append(v)
            elif g.data_ptr() != vp:
                dst.append(v)
                src.append(g)
        if zero:
            torch._foreach_zero_(zero)
        if dst:
            torch._foreach_copy_(dst, src)
        if reattach:
            self.reattach_grads()

    def reattach_grads(self):
        for (_, p), v in zip(self.entries, self._views):
            p.grad = v

    def used_grad(self):
        return self.grad[: self.n_used]


class FusedAMSGrad:
    """torch.optim.Adam(amsgrad=True) semantics as one HIP launch over FlatParams (ossid_amsgrad_step)."""

    def __init__(self, flat, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-6):
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        n = flat.n_used
        self.exp_avg = torch.zeros(n, dtype=flat.param.dtype, device=flat.param.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.max_exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.step_count = 0
        self._params = [p for _, p in flat.entries]

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def step(self):
        f = self.flat
        _lib.require_cuda(f.param)
        self.step_count += 1
        with torch.cuda.device(f.param.device):
            rc = _lib.fn("ossid_amsgrad_step")(f.param.data_ptr(), f.grad.data_ptr(), self.exp_avg.data_ptr(),
                                               self.exp_avg_sq.data_ptr(), self.max_exp_avg_sq.data_ptr(), f.n_used,
                                               self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                               self.step_count, _lib.stream())
        _lib.check(rc, "ossid_amsgrad_step")
        # the kernel wrote the parameters behind autograd's back: bump their version counters so that anything keyed on
        # them (the packed test-time plans of dtoid.Network) notices the update
        torch.autograd.graph.increment_version(self._params)
        from . import network
        network.PARAM_EPOCH[0] += 1             # (the next test-time frame checks its plans BEFORE it launches)

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "max_exp_avg_sq": self.max_exp_avg_sq}


class GradSync:
    """Keeps data-parallel replicas of the detector identical (one process per GPU; RCCL when the backend is "nccl",
    gloo in the CPU tests). Three duties, the contract being Lightning DDP's (reference train.py:93-102, whose default
    `broadcast_buffers=True` also replicates BatchNorm running statistics):

      gradients   mean over ranks of the flat gradient buffer, in a few large buckets. With `overlap=True` (default) the
                  all-reduce of a bucket is issued from autograd hooks as soon as the last gradient of the bucket has
                  been produced, so the exchange of the head's 90 MB runs under the backbone's backward pass; buckets are
                  always launched in the same (descending) order on every rank. `sync()` is the non-overlapped form
                  (used behind a hipGraph replay, which runs no hooks).
      parameters  broadcast once from rank 0 (`broadcast_params`), as DDP's constructor does.
      buffers     `sync_buffers()`: every floating-point / integer buffer of the model (BatchNorm running_mean /
                  running_var / num_batches_tracked) is broadcast from rank 0 in ONE coalesced message per dtype. Each
                  rank normalises with the statistics of its own slice of the batch (per-rank statistics, as DDP without
                  SyncBN) but every rank LEAVES a finetune step holding rank 0's running statistics -- what DDP hands to
                  all ranks at the start of the next forward. Without this the eval-mode detectors drift apart and the
                  speculative stream (stream.run_speculative) would commit frames scored by different detectors."""

    def __init__(self, flat, process_group=None, bucket_mb=32, model=None, overlap=True, force_collectives=False):
        """force_collectives: issue every collective even at world size 1 (a 1-rank RCCL group on one GPU executes every
        RCCL call, hook and stream hand-off of the multi-GPU step; tests/test_rccl_gpu.py)."""
        import torch.distributed as dist
        self.dist, self.flat, self.group, self.model = dist, flat, process_group, model
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.collectives = self.world > 1 or (bool(force_collectives) and dist.is_initialized())
        self._comm = {}
        n = flat.n_used
        per = max(1, int(bucket_mb * (1 << 20) // 4))
        self.bounds = [(s, min(n, s + per)) for s in range(0, n, per)]
        self.overlap = overlap
        self._hooks, self._works, self._armed = [], [], False
        # bucket k = the parameters whose flat range STARTS inside bounds[k] (a tensor straddling a boundary belongs to
        # the bucket it starts in; the bucket's all-reduce range is stretched to cover it)
        self._buckets = []
        if self.collectives or overlap:
            used = [(name, p) + flat.offsets[name] for name, p in flat.entries if flat.offsets[name][0] < n]
            k = 0
            cur = {"params": [], "lo": 0, "hi": 0}
            for name, p, off, cnt in used:
                while k + 1 < len(self.bounds) and off >= self.bounds[k][1]:
                    if cur["params"]:
                        self._buckets.append(cur)
                    k += 1
                    cur = {"params": [], "lo": off, "hi": off}
                if not cur["params"]:
                    cur["lo"] = off
                cur["params"].append((p, off, cnt))
                cur["hi"] = off + cnt
            if cur["params"]:
                self._buckets.append(cur)
            for b in self._buckets:
                b["views"] = [flat.grad[off:off + cnt].view_as(p) for p, off, cnt in b["params"]]

    # ---- overlapped gradient exchange ---------------------------------------------------------------------------------
    def begin(self):
        """Arm the hooks for ONE backward pass (call after flat.detach_grads(), before loss.backward())."""
        if not self._hooks:
            for bi, b in enumerate(self._buckets):
                for p, _, _ in b["params"]:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        for b in self._buckets:
            b["pending"], b["flushed"] = len(b["params"]), False
        self._next = len(self._buckets) - 1          # buckets go out last-to-first: backward produces them in that order
        self._works, self._armed = [], True
        # the stream loss.backward() is called on (begin() runs on it): where the step continues after finish()
        self._caller = torch.cuda.current_stream(self.flat.grad.device) if self.flat.grad.is_cuda else None

    def _make_hook(self, bi):
        def hook(param):
            if not self._armed:
                return
            b = self._buckets[bi]
            b["pending"] -= 1
            if b["pending"] == 0:
                self._launch_ready()
        return hook

    def _launch_ready(self):
        # fixed launch order (descending bucket index) on every rank, whatever order autograd finished them in
        while self._next >= 0 and self._buckets[self._next]["pending"] == 0:
            self._flush(self._buckets[self._next])
            self._next -= 1

    def _comm_stream(self, dev):
        """The stream a bucket is gathered and reduced on, once every stream that can hold a producer of its gradients
        has been waited for. A bucket is flushed from the post-accumulate hook of its LAST parameter, so its current
        stream is that one AccumulateGrad node's stream -- the stream the parameter was first used on in forward: main,
        or a branch slot (Network.use_train_streams) -- while the bucket mixes parameters from all of them, and
        convolution weight gradients are still in flight on the weight-gradient stream (train_ops.WGRAD_SIDE). The
        autograd engine itself only joins those streams when backward ENDS. When a hook fires, every kernel producing a
        gradient of its bucket has been ENQUEUED (the hook runs after AccumulateGrad), so waiting for what each candidate
        stream holds now is sufficient. Doing the gather + all-reduce on an own stream keeps those waits off the critical
        path: the main stream never waits for a side stream here, and the weight-gradient stream's dirty mark is left
        alone, so the end-of-backward join (train_ops.join_wgrad_stream) still happens."""
        from . import train_ops
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        comm = self._comm.get(idx)
        if comm is None:
            comm = self._comm[idx] = torch.cuda.Stream(device=idx)
        producers = [torch.cuda.current_stream(idx), self._caller]
        pool = train_ops._side_pools.get(idx) or {}
        producers += [pool.get(k) for k in ("wgrad", "b0", "b1")] + [train_ops._wg_streams.get(idx)]
        seen = set()
        for st in producers:
            if st is not None and st.cuda_stream not in seen and st.cuda_stream != comm.cuda_stream:
                seen.add(st.cuda_stream)
                comm.wait_stream(st)
        return comm

    def _flush(self, b):
        comm = self._comm_stream(self.flat.grad.device) if self.flat.grad.is_cuda else None
        dst, src = [], []
        for (p, off, cnt), v in zip(b["params"], b["views"]):
            if p.grad is None:
                dst.append(v)                     # no gradient this step: contributes zero, like gather_grads
                src.append(None)
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
        import contextlib
        with (torch.cuda.stream(comm) if comm is not None else contextlib.nullcontext()):
            for v, g_ in zip(dst, src):
                if g_ is None:
                    v.zero_()
            pairs = [(v, g_) for v, g_ in zip(dst, src) if g_ is not None]
            if pairs:
                torch._foreach_copy_([v for v, _ in pairs], [g_ for _, g_ in pairs])
                if comm is not None:
                    for _, g_ in pairs:           # freed when p.grad is re-pointed below: the allocator must not hand the
                        g_.record_stream(comm)    # memory to the producer's stream while the copy is still reading it
            if self.collectives:
                g = self.flat.grad[b["lo"]:b["hi"]]
                g.mul_(1.0 / self.world)
                if g.is_cuda and self.dist.get_backend(self.group) != "nccl":
                    torch.cuda.synchronize(g.device)          # see sync(): host-staged gloo on a shared device
                self._works.append(self.dist.all_reduce(g, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        for (p, _, _), v in zip(b["params"], b["views"]):
            p.grad = v
        b["flushed"] = True

    def finish(self):
        """After loss.backward(): flush what the hooks could not (parameters that got no gradient), wait for all
        buckets. On return flat.grad holds the mean over ranks and every p.grad is its view of the flat buffer."""
        self._armed = False
        for b in self._buckets:
            b["pending"] = 0
        self._launch_ready()
        for w in self._works:
            w.wait()
        if self.flat.grad.is_cuda:
            cur = torch.cuda.current_stream(self.flat.grad.device)
            for comm in self._comm.values():      # the gathers (and, through w.wait() above, the reductions) of every bucket
                cur.wait_stream(comm)
            if self._works and self.dist.get_backend(self.group) != "nccl":
                torch.cuda.synchronize(self.flat.grad.device)
        self._works = []

    # ---- non-overlapped form ------------------------------------------------------------------------------------------
    def sync(self):
        if not self.collectives:
            return
        g = self.flat.used_grad()
        g.mul_(1.0 / self.world)          # pre-scale once; SUM of the scaled buffers is the mean
        host_staged = g.is_cuda and self.dist.get_backend(self.group) != "nccl"
        if host_staged:
            # gloo moves device tensors through host memory and makes the stream wait on a HOST-signalled event. With
            # several ranks sharing one GPU (the rehearsal set-up, never the product one) such a device-side wait can
            # hold the hardware queue the other rank needs to get its half done: seconds per step until the scheduler
            # preempts (measured: 24 s/step). Draining the device first keeps the wait on the host.
            torch.cuda.synchronize(g.device)
        works = [self.dist.all_reduce(g[a:b], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for a, b in self.bounds]
        for w in works:
            w.wait()
        if host_staged:
            torch.cuda.synchronize(g.device)

    def broadcast_params(self, src=0):
        """Make every replica start from rank `src`'s weights AND buffers (DDP's constructor does the same)."""
        if self.collectives:
            self.dist.broadcast(self.flat.param, src=src, group=self.group)
            self.sync_buffers(src)

    def sync_buffers(self, src=0, model=None):
        """Broadcast every buffer of the model from rank `src`, coalesced per dtype (BatchNorm statistics of the whole
        detector: ~0.4 MB of floats + ~240 int64 counters -> two messages)."""
        model = self.model if model is None else model
        if not self.collectives or model is None:
            return
        by_dtype = {}
        for b in model.buffers():
            by_dtype.setdefault(b.dtype, []).append(b)
        for dt in sorted(by_dtype, key=str):
            bufs = by_dtype[dt]
            flat = torch.cat([b.detach().reshape(-1) for b in bufs])
            self.dist.broadcast(flat, src=src, group=self.group)
            torch._foreach_copy_([b.detach() for b in bufs],
                                 [c.view_as(b) for c, b in zip(flat.split([b.numel() for b in bufs]), bufs)])


_PIN_KEY = "ossid_capture_probe"


def pinned_grad_accumulators(params):
    """The parameters whose AccumulateGrad node is being KEPT ALIVE by an autograd graph of an earlier iteration (a loss or
    an output dict that still has its grad_fn, a leaked graph). Such a node carries the stream it was created on; a
    backward pass inside a hipGraph capture then has the engine record / wait events between that (non-capturing) stream
    and the capture stream, which invalidates the capture -- and ending an invalidated capture crashed the HIP runtime
    (DESIGN.md 5d). A node nothing else holds dies with our reference and a fresh one (with an empty metadata dict) is made on
    the next access; a pinned one comes back with the mark we left."""
    token, pinned = object(), []
    for p in params:
        if not p.requires_grad:
            continue
        node = p.view_as(p).grad_fn.next_functions[0][0]
        node.metadata[_PIN_KEY] = token
        del node
        node = p.view_as(p).grad_fn.next_functions[0][0]
        if node.metadata.pop(_PIN_KEY, None) is token:
            pinned.append(p)
        del node
    return pinned


class GraphedForwardBackward:
    """zero_grad + DtoidNet.forward + loss.backward() of one fixed batch shape, captured once in a hipGraph and
    replayed (the eager step is ~2 100 launches and partly host-bound). Gradients land in the FlatParams buffer, whose
    address never changes; the gradient all-reduce and the one-launch optimizer step stay outside the graph.
    BatchNorm buffers are saved and restored around the warm-up passes, so capturing changes no state.

    Capture hygiene (the round-2 `capture_end` crash): the warm-up passes run ON the capture stream and in the
    captured form of the step (one stream: no branch streams, weight gradients in line), so every autograd node the warm-up
    could leave behind belongs to the stream the capture uses; and the capture is REFUSED with a RuntimeError while any
    parameter's AccumulateGrad node is pinned by an older graph (pinned_grad_accumulators)."""

    def __init__(self, model, flat, example_batch, warmup=2):
        from . import train_ops
        self.model, self.flat = model, flat
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        dev = flat.param.device
        params = [p for _, p in flat.entries]
        pinned = pinned_grad_accumulators(params)
        if pinned:
            raise RuntimeError(
                "GraphedForwardBackward: %d parameter(s) still belong to an autograd graph of an earlier iteration (a loss / "
                "output that was kept with its grad_fn). Their AccumulateGrad nodes carry that iteration's stream into the "
                "capture and would invalidate it: drop or .detach() those tensors before capturing." % len(pinned))
        train_ops.join_wgrad_stream()                      # nothing of an earlier eager step may still be in flight
        saved = [b.detach().clone() for b in model.buffers()]
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        nets = [m for m in model.modules() if hasattr(m, "use_train_streams")]
        old_flags = [(m, m.__dict__.get("use_train_streams")) for m in nets]
        old_side = train_ops.WGRAD_SIDE
        try:
            for m in nets:
                m.use_train_streams = False
            train_ops.WGRAD_SIDE = False
            with torch.cuda.stream(cap):
                for _ in range(warmup):
                    flat.detach_grads()
                    model(self.static)["loss"].backward()
                    flat.gather_grads()
            torch.cuda.current_stream(dev).wait_stream(cap)
            with torch.no_grad():
                for b, s0 in zip(model.buffers(), saved):
                    b.copy_(s0)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=cap):
                flat.detach_grads()
                out = model(self.static)
                out["loss"].backward()
                flat.gather_grads()
                self.loss = out["loss"].detach()
                del out
        finally:
            train_ops.WGRAD_SIDE = old_side
            for m, v in old_flags:
                if v is None:
                    m.__dict__.pop("use_train_streams", None)
                else:
                    m.use_train_streams = v
        flat.zero_grad()

    def __call__(self, batch):
        for k, v in batch.items():
            if torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        return self.loss


def finetune_step(model, batch, optimizer, sync=None, graphed=None):
    """One finetune iteration on a batch already on the device; returns the detached loss. `graphed`: a
    GraphedForwardBackward built for this batch shape (optional). With `sync` (data parallel): gradient mean over ranks
    (overlapped with backward in the eager form), then the step, then rank 0's BatchNorm buffers to every rank."""
    if graphed is not None:
        loss = graphed(batch)
        if sync is not None:
            sync.sync()
        optimizer.step()
    else:
        loss = _finetune_step_eager(model, batch, optimizer, sync)
    if sync is not None:
        sync.sync_buffers(model=model if sync.model is None else None)
    return loss


def _finetune_step_eager(model, batch, optimizer, sync=None):
    """One finetune iteration on a batch already on the device; returns the detached loss."""
    out = model(batch)
    loss = out["loss"]
    flat = getattr(optimizer, "flat", None)
    if flat is None and sync is not None:
        flat = sync.flat               # a torch optimizer over FlatParams views: same gather protocol
    if flat is not None:               # gradients gathered into the flat buffer by multi-tensor copies
        flat.detach_grads(collect=sync is None and hasattr(optimizer, "flat"))
        if sync is not None and sync.overlap:
            sync.begin()               # per-bucket gather + all-reduce from autograd hooks, under the backward pass
            loss.backward()
            sync.finish()
        else:
            loss.backward()
            if sync is None and hasattr(optimizer, "flat"):
                flat.gather_grads(reattach=False)     # the fused step reads the flat buffer, not p.grad: launch it first
                optimizer.step()
                flat.reattach_grads()
                return loss.detach()
            flat.gather_grads()
            if sync is not None:
                sync.sync()
    else:
        optimizer.zero_grad()
        loss.backward()
    optimizer.step()
    return loss.detach()


def finetuneDtoid(model, train_set, optimizer, batch_size=8, epochs=5, num_workers=0, collate_fn=None, sync=None,
                  sampler=None):
    """Counterpart of online_learning.py:650-679: `epochs` passes over `train_set` in shuffled batches, BatchNorm in
    training mode, then back to eval. `optimizer` may be a torch optimizer (the reference's) or FusedAMSGrad."""
    from torch.utils.data import DataLoader
    loader = DataLoader(train_set, batch_size=batch_size, num_workers=num_workers, shuffle=sampler is None,
                        sampler=sampler, pin_memory=True, collate_fn=collate_fn)
    dev = next(model.parameters()).device
    model.train()
    losses = []
    for _ in range(epochs):
        for batch in loader:
            batch = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
            losses.append(finetune_step(model, batch, optimizer, sync))
    model.eval()
    return [float(v) for v in losses]
