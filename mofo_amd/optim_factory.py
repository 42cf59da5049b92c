"""Drop-in for the AdamW path of the reference's ``optim_factory.py`` (create_optimizer / get_parameter_groups,
optim_factory.py:49-127): same two parameter groups (decayed: >=2-D weights; not decayed: 1-D tensors, ``.bias`` and the
model's ``no_weight_decay()`` names), each carrying ``lr_scale`` -- but the update itself is ONE fused HIP kernel over
the model's flat parameter / gradient / moment buffers, which also refreshes the bf16 copies the GEMMs read.
The other 20 optimizers of the reference factory are not used by the pretraining recipe (PRETRAIN.md:25-26) and are
out of scope; asking for one raises."""
import torch

from . import ops


def get_parameter_groups(model, weight_decay=1e-5, skip_list=(), get_num_layer=None, get_layer_scale=None):
    """optim_factory.py:49-88 (the layer-decay branch belongs to fine-tuning and is not built)."""
    if get_num_layer is not None or get_layer_scale is not None:
        raise NotImplementedError("layer-wise lr decay is a fine-tuning feature (out of scope)")
    groups = {"decay": {"weight_decay": weight_decay, "params": [], "lr_scale": 1.},
              "no_decay": {"weight_decay": 0., "params": [], "lr_scale": 1.}}
    names = {"decay": [], "no_decay": []}
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        g = "no_decay" if (len(param.shape) == 1 or name.endswith(".bias") or name in skip_list) else "decay"
        groups[g]["params"].append(param)
        names[g].append(name)
    # reference order of creation = order of first appearance in named_parameters()
    first = next(iter(model.named_parameters()))[0]
    order = ["no_decay", "decay"] if first in names["no_decay"] else ["decay", "no_decay"]
    return [groups[k] for k in order if groups[k]["params"]], {k: names[k] for k in order}


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled decay, bias correction, eps outside the sqrt) on flat buffers.

    ``param_groups`` keeps the reference's keys (lr, weight_decay, lr_scale, betas, eps) so the engine's per-step
    schedule writes (engine_for_pretraining.py:31-37) work unchanged.  ``grad_norm`` / ``max_norm`` let the caller fuse
    clip_grad_norm_ (utils.py:359) into the same kernel without a host sync."""

    def __init__(self, model, param_groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(param_groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        self._store = None
        self._step = 0
        self.exp_avg = None
        self.exp_avg_sq = None
        self.measure_exposed = False      # bench.py: event pairs around the waits for the data-parallel gradient ranges
        self.exposed_events = []          # one list of (start, end) events per step while measure_exposed is on
        self._bind()

    def _bind(self):
        rt = self.model.runtime()
        st = rt.store
        if self._store is not st:
            if self._store is not None and self._step > 0:
                raise RuntimeError("the model's parameters were moved after optimisation started; optimizer state would be lost")
            self._store = st
            self.exp_avg = torch.zeros_like(st.params)
            self.exp_avg_sq = torch.zeros_like(st.params)
            self._verify_groups(st)
        return rt, st

    def _verify_groups(self, st):
        """The kernel picks decay / no-decay per 1024-element chunk from the flat store's table (built from the model's
        no_weight_decay() and the reference's rule, optim_factory.py:56-61) and reads ONE betas / eps: param_groups that
        say something else (a custom skip_list, filter_bias_and_bn=False with a non-zero decay on biases, per-group betas)
        would be reported but not applied -- refuse them instead."""
        if len(self.param_groups) > 2:
            raise NotImplementedError("the fused AdamW applies two groups (decay / no decay), got %d" % len(self.param_groups))
        g0 = self.param_groups[0]
        for g in self.param_groups[1:]:
            if tuple(g["betas"]) != tuple(g0["betas"]) or g["eps"] != g0["eps"]:
                raise NotImplementedError("the fused AdamW applies one betas / eps to every group")
        if not all("_decayed" in g for g in self.param_groups):
            return      # groups restored from a foreign checkpoint are tagged in load_state_dict
        names = {id(p): n for n, p in self.model.named_parameters()}
        table = st.chunk_group.cpu().numpy()
        for g in self.param_groups:
            want = 0 if g["_decayed"] else 1
            for p in g["params"]:
                n = names.get(id(p))
                if n is None:
                    raise ValueError("optimizer parameter is not a parameter of the bound model")
                o = st.offset[n]
                if int(table[o // 1024]) != want:
                    raise NotImplementedError(f"{n}: param_groups put it in the {'decay' if g['_decayed'] else 'no-decay'} group but the "
                                              "flat store's chunk table (model.no_weight_decay() + the reference's 1-D / .bias rule) says "
                                              "otherwise; the fused kernel would apply the table")

    def zero_grad(self, set_to_none: bool = False):
        _, st = self._bind()
        st.zero_grads()
        if not st.grads_attached():
            st.attach_grads()

    @torch.no_grad()
    def undo_skipped_step(self):
        """the last step() was declined on the device (non-finite loss / status word): take its count back"""
        if self._step > 0:
            self._step -= 1

    def step(self, closure=None, grad_norm=None, max_norm=0.0, norm_out=None, ranges=None):
        """``norm_out`` (device f32 [1], only without clipping): the update pass also leaves the global gradient L2 norm
        there -- the norm is reported, not needed before the update, so the gradients are read once instead of twice.
        ``ranges`` (data parallelism, un-clipped step): iterable of ``(lo, hi, wait)`` tiling the flat buffers in the order
        their gradient all-reduces were issued; ``wait()`` makes the stream wait for that range's exchange, then the range
        is updated -- the HBM-bound update of the early ranges runs while the last range is still on the wire.
        Gate: the device-side gate (mofo_adamw_gated) can decline the update AFTER this call has advanced the host-side step
        counter; a caller that keeps training past a skipped step (the drop-in engine does not: it exits on the non-finite
        loss like engine_for_pretraining.py:168-170) calls ``undo_skipped_step()`` so the bias correction stays in step.  The
        gate reflects the last backward only: under gradient accumulation, check every micro-batch's loss on the host."""
        if closure is not None:
            raise NotImplementedError("closure")
        rt, st = self._bind()
        g0 = next(g for g in self.param_groups if g["_decayed"]) if any(g.get("_decayed") for g in self.param_groups) else self.param_groups[0]
        g1 = next((g for g in self.param_groups if not g["_decayed"]), g0)
        self._step += 1
        b1, b2 = g0["betas"]
        partial = None
        if norm_out is not None:
            if max_norm:
                raise ValueError("norm_out is for the un-clipped step (clipping needs the norm before the update)")
            if getattr(self, "_norm_partial", None) is None or self._norm_partial.device != st.params.device:
                self._norm_partial = torch.empty(2048, dtype=torch.float32, device=st.params.device)
            partial = self._norm_partial
        hyper = (float(g0["lr"]), float(g0["weight_decay"]), float(g1["lr"]), float(g1["weight_decay"]), float(b1), float(b2), float(g0["eps"]), self._step)
        # device-side gate of this update (set by the backward that produced the gradients; consumed once)
        gf, gz, go = getattr(rt, "step_gate", None) or (None, None, None)
        rt.step_gate = None
        gate = dict(gate_finite=gf, gate_zero=gz, gate_one=go)
        # MOFO_FP8=1: this pass also writes the e4m3 shadow of the fp8 forward's weights (delayed per-matrix scale = 448 / the maximum the
        # PREVIOUS update left; ops.fp8_roll_scales).  Only while that shadow is current -- after a load_state_dict / a foreign write the
        # next forward re-quantises from the bf16 shadow with exact scales and this path resumes with the update after it.
        # ... and only while nobody has written the masters since the bf16 shadow was made (the version counters, not just the epochs:
        # a load_state_dict or manual write between the last forward and this step would otherwise be quantised with the OLD weights'
        # delayed scale -- no margin, saturating at 448 -- and marked fresh, so the next forward would run on a mis-scaled shadow)
        q8 = hasattr(st, "shadow8") and st.shadow8_current() and st._version() == st._shadow_version
        if q8:
            ops.fp8_roll_scales(st._amax_ws, st.w_scale, st.w_scale_inv, **gate)
        if ranges is not None:
            if max_norm or grad_norm is not None:
                raise ValueError("range-by-range update is for the un-clipped step")
            ranges = list(ranges)
            slots = sum(ops.adamw_blocks(hi - lo) for lo, hi, _ in ranges)
            if getattr(self, "_range_partial", None) is None or self._range_partial.numel() < slots:
                self._range_partial = torch.empty(max(slots, 1), dtype=torch.float32, device=st.params.device)
            covered, slot = 0, 0
            measure = getattr(self, "measure_exposed", False)
            pairs = []
            for lo, hi, wait in ranges:
                if lo % 1024 or hi % 1024 or hi <= lo:
                    raise ValueError(f"range [{lo}, {hi}) is not a 1024-aligned slice of the flat buffers")
                if measure:     # bench.py: how long the compute stream idles for this range's exchange (0 when it had landed)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    wait()
                    e1.record()
                    pairs.append((e0, e1))
                else:
                    wait()
                nb = ops.adamw_blocks(hi - lo)
                ops.adamw(st.params[lo:hi], st.grads[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], st.shadow[lo:hi],
                          st.chunk_group[lo // 1024:hi // 1024], *hyper, norm_partial=self._range_partial[slot:slot + nb] if norm_out is not None else None,
                          q8=st.q8_args(lo, hi) if q8 else None, **gate)
                slot += nb
                covered += hi - lo
            if measure:
                self.exposed_events.append(pairs)
            if covered != st.params.numel():
                raise RuntimeError(f"ranges cover {covered} of {st.params.numel()} elements")
            if norm_out is not None:
                ops.norm_finalize(self._range_partial, slot, norm_out)
        else:
            ops.adamw(st.params, st.grads, self.exp_avg, self.exp_avg_sq, st.shadow, st.chunk_group, *hyper, grad_norm=grad_norm,
                      max_norm=float(max_norm) if max_norm else 0.0, norm_partial=partial, norm_out=norm_out, q8=st.q8_args() if q8 else None, **gate)
        st.mark_shadow_fresh()
        if q8:
            st.mark_shadow8_fresh()

    # checkpoint.  Written in torch.optim.AdamW's own state_dict layout -- state[i] = {'step','exp_avg','exp_avg_sq'} with i
    # counting parameters group by group, param_groups[g]['params'] = index lists -- i.e. exactly what the reference's
    # utils.save_model (utils.py:417-423) stores under 'optimizer', so checkpoints interchange in both directions.
    def _indexed_params(self):
        names = {id(p): n for n, p in self.model.named_parameters()}
        out, i = [], 0
        for g in self.param_groups:
            for p in g["params"]:
                out.append((i, names[id(p)], p))
                i += 1
        return out

    def state_dict(self):
        _, st = self._bind()
        state = {}
        for i, name, p in self._indexed_params():
            o, n = st.offset[name], p.numel()
            state[i] = {"step": torch.tensor(float(self._step)), "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        groups, i = [], 0
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            d["params"] = list(range(i, i + len(g["params"])))
            i += len(g["params"])
            groups.append(d)
        return {"state": state if self._step > 0 else {}, "param_groups": groups}

    def load_state_dict(self, sd):
        _, st = self._bind()
        idx = self._indexed_params()
        if len(sd["param_groups"]) != len(self.param_groups) or sum(len(g["params"]) for g in sd["param_groups"]) != len(idx):
            raise ValueError("optimizer state does not match this model's parameter groups")
        state = sd["state"]
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = set()
        for i, name, p in idx:
            if i not in state:
                continue
            e = state[i]
            o, n = st.offset[name], p.numel()
            if tuple(e["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state shape mismatch for {name}")
            self.exp_avg[o:o + n].copy_(e["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(e["exp_avg_sq"].reshape(-1))
            steps.add(int(float(e["step"])))
        if len(steps) > 1:
            raise ValueError("parameters carry different step counts; the fused update keeps one")
        self._step = steps.pop() if steps else 0
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in saved.items() if k != "params"})
            g.setdefault("_decayed", g.get("weight_decay", 0) > 0)


def create_optimizer(args, model, get_num_layer=None, get_layer_scale=None, filter_bias_and_bn=True, skip_list=None):
    """optim_factory.py:91-175 for ``--opt adamw`` (the pretraining recipe)."""
    opt_lower = args.opt.lower().split('_')[-1]
    if opt_lower != 'adamw':
        raise NotImplementedError(f"optimizer '{args.opt}': only adamw (the pretraining recipe, PRETRAIN.md:25) is built")
    model = getattr(model, "module", model)
    weight_decay = args.weight_decay
    skip = skip_list if skip_list is not None else (model.no_weight_decay() if hasattr(model, 'no_weight_decay') else {})
    groups, names = get_parameter_groups(model, weight_decay, skip, get_num_layer, get_layer_scale)
    if not (weight_decay and filter_bias_and_bn):
        for g in groups:
            g["weight_decay"] = weight_decay or 0.
    for g, k in zip(groups, names):
        g["_decayed"] = (k == "decay")
    kw = dict(lr=args.lr, weight_decay=0.)
    if getattr(args, 'opt_eps', None) is not None:
        kw['eps'] = args.opt_eps
    if getattr(args, 'opt_betas', None) is not None:
        kw['betas'] = tuple(args.opt_betas)
    return FusedAdamW(model, groups, **kw)
