"""Fused global-norm clip + Adam over the flat parameter buffer (HIP), with the torch.optim interface the reference
loop expects (train/train_artiboost.py:91-96; anakin/utils/netutils.py:26-33)."""
import torch

from . import _lib as L


class FusedClipAdam(torch.optim.Optimizer):
    """Adam(lr, betas, eps, weight_decay=0) on ONE flat fp32 parameter whose .grad is the flat gradient buffer.
    `max_norm` folds clip_grad_norm_ into the same pass; when the caller already clipped (the reference loop calls
    torch.nn.utils.clip_grad_norm_ itself) construct with max_norm=None."""

    def __init__(self, params, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=None, model=None):
        if weight_decay:
            raise NotImplementedError("weight_decay (the reference trains with WEIGHT_DECAY: 0)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.max_norm = max_norm
        self.model = model
        self._part = None
        self.total_norm = None
        self.hyper = None           # device float[3]; set by use_device_hyper() for hipGraph replay
        self._hyper_host = None
        self.graph_steps = 0
        self._eager_hyper = None

    HYPER_SLOTS = 32

    def use_device_hyper(self, device):
        """Device-side {lr, bias corrections} for hipGraph replay.  The host runs many replayed steps ahead of the device
        (nothing in an epoch synchronises), so the pinned staging is a RING: slot k is rewritten only after the copy that
        last read it has executed (its event) -- a single reused pinned tensor could be overwritten by step i+k's values
        before step i's queued H2D copy ran."""
        self.hyper = torch.zeros(3, dtype=torch.float32, device=device)
        self._hyper_host = torch.zeros((self.HYPER_SLOTS, 3), dtype=torch.float32).pin_memory()
        self._hyper_events = [None] * self.HYPER_SLOTS
        self._hyper_next = 0

    def advance_hyper(self):
        """Host side of a replayed step: bump the step count and push {lr, bias corrections} to the device."""
        self.graph_steps += 1
        for st in self.state.values():          # keep the eager step count in line (state_dict / a later eager step)
            st["step"] = self.graph_steps
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        k = self._hyper_next
        self._hyper_next = (k + 1) % self.HYPER_SLOTS
        if self._hyper_events[k] is not None:
            self._hyper_events[k].synchronize()          # the copy that read this slot HYPER_SLOTS steps ago has run
        slot = self._hyper_host[k]
        slot[0] = g["lr"]
        slot[1] = 1.0 - b1 ** self.graph_steps
        slot[2] = (1.0 - b2 ** self.graph_steps) ** 0.5
        self.hyper.copy_(slot, non_blocking=True)
        ev = self._hyper_events[k] or torch.cuda.Event()
        ev.record()
        self._hyper_events[k] = ev

    def state_dict(self):
        """torch.optim state_dict (one flat parameter: exp_avg / exp_avg_sq are the flat Adam moments) + the replay step
        count that drives the bias corrections under hipGraph replay."""
        sd = super().state_dict()
        sd["graph_steps"] = self.graph_steps
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        self.graph_steps = int(state_dict.pop("graph_steps", 0))
        super().load_state_dict(state_dict)
        for st in self.state.values():
            self.graph_steps = max(self.graph_steps, int(st.get("step", 0)))

    def grad_norm(self, grad):
        if self._part is None:
            self._part = torch.empty(1024, dtype=torch.float32, device=grad.device)
            self.total_norm = torch.empty(1, dtype=torch.float32, device=grad.device)
        L.check(L.lib().ab_grad_norm(L.ptr(grad), L.l(grad.numel()), L.ptr(self._part), L.ptr(self.total_norm),
                                     L.stream()), "ab_grad_norm")
        return self.total_norm

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                g = p.grad
                tn = self.grad_norm(g) if self.max_norm else None
                lp = lp_planes = None
                net = self.model.net if (self.model is not None and p is self.model.flat_param) else None
                if net is not None and getattr(net, "x3", False):
                    if net.lp is None or tuple(net.lp.shape) != (2, p.numel()):
                        net.lp = torch.empty((2, p.numel()), dtype=torch.bfloat16, device=p.device)
                    lp_planes = net.lp               # split-bf16 weight planes, refreshed by the Adam pass itself
                elif net is not None and net.dtype == torch.bfloat16:
                    if net.lp is None or net.lp.dtype != torch.bfloat16:
                        net.lp = torch.empty(p.numel(), dtype=torch.bfloat16, device=p.device)
                    lp = net.lp
                b1, b2 = group["betas"]
                hyper = self.hyper
                if hyper is None:        # eager step: the same host-computed {lr, bias corrections} a replayed step reads, so
                    if self._eager_hyper is None or self._eager_hyper.device != p.device:      # both modes update identically
                        self._eager_hyper = torch.zeros(3, dtype=torch.float32, device=p.device)
                    hyper = self._eager_hyper
                    hyper.copy_(torch.tensor([group["lr"], 1.0 - b1 ** st["step"], (1.0 - b2 ** st["step"]) ** 0.5], dtype=torch.float32))
                if lp_planes is not None:
                    L.check(L.lib().ab_clip_adam_x3(L.ptr(p), L.ptr(g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]),
                                                    L.l(p.numel()), L.ptr(tn), L.f(self.max_norm or 0.0), L.f(group["lr"]),
                                                    L.f(b1), L.f(b2), L.f(group["eps"]), L.i(st["step"]), L.ptr(hyper),
                                                    L.ptr(lp_planes[0]), L.ptr(lp_planes[1]), L.stream()), "ab_clip_adam_x3")
                else:
                    L.check(L.lib().ab_clip_adam(L.ptr(p), L.ptr(g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]),
                                                 L.l(p.numel()), L.ptr(tn), L.f(self.max_norm or 0.0), L.f(group["lr"]),
                                                 L.f(b1), L.f(b2), L.f(group["eps"]), L.i(st["step"]), L.ptr(hyper),
                                                 L.ptr(lp), L.stream()), "ab_clip_adam")
                if net is not None:
                    net.refresh_after_update(lp_fresh=lp is not None or lp_planes is not None)
        return None
