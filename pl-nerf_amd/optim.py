"""Adam as the reference configures it (run_plnerf.py:446-447: lr, betas (0.9, 0.999), eps 1e-8, no weight
decay, no amsgrad), stepping a whole parameter group with ONE launch of plnerf_adam_step.

`FlatAdam` is a torch.optim.Adam: same constructor arguments, same `param_groups` (the training loop rewrites
`group['lr']` every iteration, run_plnerf.py:1311-1315), same `state_dict()` layout -- so the checkpoint the
reference writes (run_plnerf.py:1324-1332, `optimizer_state_dict`) loads here and vice versa.  What differs is
the storage: at construction the group's parameters are re-homed as consecutive slices of one flat fp32 buffer
(their `.data` become views; `load_state_dict` / `copy_` keep working), and so are `exp_avg` / `exp_avg_sq`.
functional.MlpFn.backward hands autograd the network's 24 gradients as slices of one buffer in the same order,
so a step is one elementwise kernel over four flat arrays per network instead of a multi-tensor launch over 24
tensor lists; gradients in any other layout are stepped run by run (worst case one launch per tensor).
Parameters without a gradient are skipped, as in torch.  CPU parameters fall back to torch's own step.
"""
import torch

from . import _lib as L


def contiguous_runs(tensors):
    """Split a list of fp32 tensors into maximal runs that are consecutive slices, in order, of one buffer.
    Returns [(first index, one-past-last index, flat view over the run)]; a tensor that is not contiguous fp32
    forms a run of its own with view None."""
    runs = []
    i, n = 0, len(tensors)
    while i < n:
        t0 = tensors[i]
        if t0.dtype != torch.float32 or not t0.is_contiguous():
            runs.append((i, i + 1, None))
            i += 1
            continue
        store_end = t0.untyped_storage().data_ptr() + t0.untyped_storage().nbytes()
        expect = t0.data_ptr() + t0.numel() * 4
        total = t0.numel()
        j = i + 1
        while j < n:
            t = tensors[j]
            if (t.dtype != torch.float32 or not t.is_contiguous() or t.device != t0.device or t.data_ptr() != expect
                    or expect + t.numel() * 4 > store_end):     # (neighbouring allocations that merely touch)
                break
            expect += t.numel() * 4
            total += t.numel()
            j += 1
        runs.append((i, j, t0.as_strided((total,), (1,))))
        i = j
    return runs


def flat_view_of(tensors):
    """One flat view over `tensors` if they are consecutive slices, in order, of one buffer; else None."""
    if not tensors or any(t is None for t in tensors):
        return None
    runs = contiguous_runs(tensors)
    return runs[0][2] if len(runs) == 1 else None


class FlatAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, guards=None):
        """guards: up to two networks (objects with `.status_word()`: nerf.NeRF) or 1-element int32 device tensors;
        while the range status word of any of them is non-zero the step kernels leave parameters and moments untouched
        (a step whose forward left the half range must not reach the weights) and count the launch in a device word;
        `withheld_steps()` -- called by the train steps' check_range() polling -- takes those steps back out of the
        host-side step counts, so that the bias corrections do not drift.  Networks are resolved to their status word
        at step time (the packed buffer a word lives in is re-allocated when a network changes device or precision)."""
        super().__init__(params, lr=lr, betas=betas, eps=eps)
        self.guards = list(guards or [])
        if len(self.guards) > 2:
            raise ValueError("FlatAdam: at most two guard words per optimizer (plnerf_adam_step takes two)")
        self._flat = []          # per group: dict(param, m, v, grad, step) or None (torch's own step)
        self._withheld = None    # device counter of guarded launches that changed nothing
        self._launches_per_step = 1
        for group in self.param_groups:
            ps = group['params']
            ok = len(ps) > 0 and all(p.is_cuda and p.dtype == torch.float32 and p.device == ps[0].device for p in ps)
            self._flat.append(self._flatten(ps) if ok else None)

    def _flatten(self, ps):
        sizes = [p.numel() for p in ps]
        dev = ps[0].device
        # parameters that already are consecutive slices of one buffer (another FlatAdam over the same network: the
        # reference's single-pass configuration steps TWO Adams over the coarse weights, run_plnerf.py:438-447) stay
        # where they are -- re-homing them would orphan the other optimizer's buffer
        flat = flat_view_of([p.data for p in ps])
        adopt = flat is not None
        if not adopt:
            flat = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        m = torch.zeros_like(flat)
        v = torch.zeros_like(flat)
        for p, fp, fm, fv in zip(ps, flat.split(sizes), m.split(sizes), v.split(sizes)):
            if not adopt:
                fp = fp.view(p.shape)
                fp.copy_(p.data)
                p.data = fp
            # torch's own state layout, so that state_dict() is the reference's optimizer_state_dict
            self.state[p] = {'step': torch.tensor(0.0), 'exp_avg': fm.view(p.shape), 'exp_avg_sq': fv.view(p.shape)}
        return {'param': flat, 'm': m, 'v': v, 'sizes': sizes}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)      # replaces the state tensors: move them back into the flat buffers
        for group, fl in zip(self.param_groups, self._flat):
            if fl is None:
                continue
            for p, fm, fv in zip(group['params'], fl['m'].split(fl['sizes']), fl['v'].split(fl['sizes'])):
                st = self.state.get(p)
                if not st:
                    self.state[p] = {'step': torch.tensor(0.0), 'exp_avg': fm.view(p.shape).zero_(),
                                     'exp_avg_sq': fv.view(p.shape).zero_()}
                    continue
                fm.view(p.shape).copy_(st['exp_avg'])
                fv.view(p.shape).copy_(st['exp_avg_sq'])
                st['exp_avg'], st['exp_avg_sq'] = fm.view(p.shape), fv.view(p.shape)
                st['step'] = torch.as_tensor(float(st['step']), dtype=torch.float32)
            fl.pop('uniform_step', None)

    @staticmethod
    def _uniform_grads(ps, base, offs):
        """Every gradient sits at its parameter's offset behind `base` (24 pointer compares: the cheap confirmation of what
        the first / last check suggests, without contiguous_runs' storage queries)."""
        for k, p in enumerate(ps):
            if p.grad.data_ptr() != base + 4 * offs[k]:
                return False
        return True

    def _uniform_steps(self, fl, ps, s0):
        """EVERY parameter's step count equals s0 (the one launch uses a single bias correction for all 24 tensors).  The
        full comparison runs whenever the count is not the one this optimizer itself left behind at its last one-launch
        step (a partially loaded state_dict, a parameter that skipped steps for lack of a gradient, wound-back counts)."""
        if fl.get('uniform_step') == s0:
            return True
        return all(float(self.state[p]['step']) == s0 for p in ps)

    def _guard_ptrs(self, guards=None):
        """The (up to two) words a launch is guarded by, resolved NOW: a network's packed buffer -- and with it its status
        word -- is re-allocated when the network moves or changes precision, a cached view would watch dead memory.  A
        word is any 4-byte device value whose non-zero BITS mean "withhold": a network's int32 status word, or the float
        a data-parallel exchange summed over the ranks' words (dp.GradientBucket.tails)."""
        ptrs = []
        for g in (self.guards if guards is None else guards):
            word = g.status_word() if hasattr(g, "status_word") else g
            ptrs.append(L.dptr(word, "guard", word.dtype if word.dtype in (torch.int32, torch.float32) else torch.int32))
        if len(ptrs) > 2:
            raise ValueError("FlatAdam: at most two guard words per step (plnerf_adam_step takes two)")
        return (ptrs + [None, None])[:2]

    def _withheld_ptr(self, dev):
        if not (self.guards or getattr(self, "_guarded_now", False)):
            return None
        if self._withheld is None or self._withheld.device != dev:
            self._withheld = torch.zeros(1, device=dev, dtype=torch.int32)
        return L.dptr(self._withheld, "withheld", torch.int32)

    def withheld_steps(self, reset=True):
        """Steps the guarded kernels withheld since the last call (synchronises: one 4-byte read).  With `reset` the
        host-side step counts -- advanced before each launch, because the host cannot know what the device decided --
        are wound back by that many, so Adam's bias corrections are those of the steps that really happened."""
        if self._withheld is None:
            return 0
        n = int(self._withheld.item()) // max(self._launches_per_step, 1)
        if n and reset:
            self._withheld.zero_()
            for fl in self._flat:
                if fl is not None:
                    fl.pop('uniform_step', None)
            for group in self.param_groups:
                for p in group['params']:
                    st = self.state.get(p)
                    if st and 'step' in st:
                        st['step'].sub_(float(n)).clamp_(min=0.0)
        return n

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, clip_value=0.0, guards=None):
        """grad_scale: every gradient entry is multiplied by it inside the step kernel (a data-parallel caller passes
        1 / world for gradients that hold the SUM over the ranks).  guards: this call's guard words instead of the
        constructor's (a data-parallel step is guarded by the ranks' summed status, dp.GradientBucket.tails).
        clip_value > 0: every gradient entry is clamped to [-clip_value, clip_value] inside the step kernel (what
        torch.nn.utils.clip_grad_value_ between backward and step does, run_nerf_sample_based_depth.py:1156 -- minus
        its 48 launches; `.grad` itself keeps the unclipped values)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if any(fl is None for fl in self._flat):
            if grad_scale != 1.0:
                raise RuntimeError("FlatAdam: grad_scale needs flat (GPU fp32) parameter groups")
            if clip_value > 0.0:
                torch.nn.utils.clip_grad_value_([p for g in self.param_groups for p in g['params']], clip_value)
            super().step(closure=None)      # (the closure, if any, was evaluated above)
            return loss
        guard_a, guard_b = self._guard_ptrs(guards)
        self._guarded_now = bool(self.guards if guards is None else guards)
        launches = 0
        for group, fl in zip(self.param_groups, self._flat):
            ps = group['params']
            if all(p.grad is None for p in ps):
                continue
            if group.get('weight_decay', 0) or group.get('amsgrad') or group.get('maximize'):
                raise RuntimeError("FlatAdam implements the reference's Adam: no weight decay / amsgrad / maximize")
            offs = fl.get('offs')
            if offs is None:
                offs = [0]
                for n in fl['sizes']:
                    offs.append(offs[-1] + n)
                fl['offs'] = offs
            # the parameters must still be the slices of this optimizer's buffer.  First and last slice are checked on every
            # step (two pointer reads); the full walk over all of them when those do not line up
            base = fl['param'].data_ptr()
            in_place = ps[0].data.data_ptr() == base and ps[-1].data.data_ptr() == base + 4 * offs[-2]
            if not in_place:
                view = flat_view_of([p.data for p in ps])
                in_place = view is not None and view.data_ptr() == base
            if not in_place:
                # not this optimizer's buffer any more: .to(device) after construction, or a second FlatAdam built over
                # the same parameters re-homed them into ITS buffer -- stepping fl['param'] would update dead memory
                raise RuntimeError("FlatAdam: the parameters no longer live in this optimizer's flat buffer (they were "
                                   "re-allocated, or another FlatAdam was built over them afterwards); build one "
                                   "optimizer per parameter set, after .to(device)")
            # parameters without a gradient are skipped, as torch.optim.Adam does; the others are stepped run by
            # run: a run = consecutive parameters whose gradients are consecutive slices of one buffer and whose
            # step counts agree (one network's backward = one run)
            live = [k for k, p in enumerate(ps) if p.grad is not None]
            b1, b2 = group['betas']
            k = 0
            # the common case in one look: every parameter has a gradient, the gradients are the slices of ONE buffer in
            # parameter order (functional.MlpFn.backward's layout) and the step counts agree -> one launch, no walk
            if len(live) == len(ps) and len(ps) > 1:
                g0, g1 = ps[0].grad, ps[-1].grad
                s0, s1 = self.state[ps[0]]['step'], self.state[ps[-1]]['step']
                if (g0.dtype == torch.float32 and g1.data_ptr() == g0.data_ptr() + 4 * offs[-2] and
                        g0.untyped_storage().data_ptr() == g1.untyped_storage().data_ptr() and
                        g0.is_contiguous() and g1.is_contiguous() and float(s0) == float(s1) and
                        self._uniform_steps(fl, ps, float(s0)) and self._uniform_grads(ps, g0.data_ptr(), offs)):
                    steps = [self.state[p]['step'] for p in ps]
                    torch._foreach_add_(steps, 1.0)
                    fl['uniform_step'] = float(s0) + 1.0
                    L.check(L.lib().plnerf_adam_step(
                        L.dptr(fl['param']), L.dptr(g0.as_strided((offs[-1],), (1,))), L.dptr(fl['m']), L.dptr(fl['v']),
                        offs[-1], float(group['lr']), float(b1), float(b2), float(group['eps']),
                        int(float(steps[0])), float(grad_scale), float(clip_value), guard_a, guard_b,
                        self._withheld_ptr(fl['param'].device), L.stream()), "plnerf_adam_step")
                    launches += 1
                    k = len(live)
            while k < len(live):
                e = k + 1
                while e < len(live) and live[e] == live[e - 1] + 1 and \
                        float(self.state[ps[live[e]]]['step']) == float(self.state[ps[live[k]]]['step']):
                    e += 1
                idx = live[k:e]
                for a, b, gview in contiguous_runs([ps[q].grad for q in idx]):
                    first, last = idx[a], idx[b - 1]
                    if gview is None:
                        gview = ps[first].grad.to(torch.float32).contiguous().reshape(-1)
                    lo, hi = offs[first], offs[last + 1]
                    steps = [self.state[ps[q]]['step'] for q in idx[a:b]]
                    torch._foreach_add_(steps, 1.0)
                    fl.pop('uniform_step', None)
                    L.check(L.lib().plnerf_adam_step(
                        L.dptr(fl['param'][lo:hi]), L.dptr(gview), L.dptr(fl['m'][lo:hi]), L.dptr(fl['v'][lo:hi]),
                        hi - lo, float(group['lr']), float(b1), float(b2), float(group['eps']),
                        int(steps[0].item()), float(grad_scale), float(clip_value), guard_a, guard_b,
                        self._withheld_ptr(fl['param'].device), L.stream()), "plnerf_adam_step")
                    launches += 1
                k = e
        self._launches_per_step = max(launches, 1)
        return loss
