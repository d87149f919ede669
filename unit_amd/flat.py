"""Flat parameter / gradient / momentum storage (288 GB of HBM: keep everything resident, contiguous and fusable).

All trainable parameters of a model live in ONE fp32 device buffer laid out in reverse-backward order (the order in
which their gradients become final), with the matching gradient and momentum buffers. Consequences:
  * `param.data` / `param.grad` are views (conv weights are channels_last views, i.e. physically [K][R][S][C] -- the
    layout the HIP wgrad kernel writes), so wgrad output IS the .grad tensor, no copies;
  * SGD is one fused kernel launch per hyper-parameter segment over contiguous memory (unit_sgd_momentum);
  * data-parallel gradient buckets are contiguous slices -> zero-copy RCCL all-reduce (unit_amd/parallel.py);
  * groups of small Linear/1x1 heads that are evaluated as ONE GEMM are adjacent rows of one fused weight matrix.
State-dict keys and logical shapes stay exactly the reference's (checkpoints load with load_state_dict).
"""
import torch


def _align(n, a):
    return (n + a - 1) // a * a


class FlatStore:
    ALIGN = 64  # elements (256 B)

    def __init__(self, device):
        self.device = device
        self.entries = []   # dict(name, param, offset, numel, group)
        self.size = 0
        self.params = None
        self.grads = None
        self.momentum = None
        self._first_step = True

    def add(self, name, param, pad_after=True):
        """Registers a parameter at the current offset; pad_after=False keeps the next one adjacent (fused heads)."""
        self.entries.append(dict(name=name, param=param, offset=self.size, numel=param.numel()))
        self.size += param.numel()
        if pad_after:
            self.size = _align(self.size, self.ALIGN)

    def reserve(self, numel):
        """zero-initialised padding rows that belong to a fused matrix (e.g. rows 101..103 of a 104-row head)."""
        off = self.size
        self.size = _align(self.size + numel, self.ALIGN)
        return off

    def pad(self):
        self.size = _align(self.size, self.ALIGN)

    @staticmethod
    def _view(flat, off, p):
        n = p.numel()
        seg = flat[off:off + n]
        if p.dim() == 4 and not getattr(p, "_unit_plain_layout", False):
            k, c, r, s = p.shape
            return seg.view(k, r, s, c).permute(0, 3, 1, 2)   # logical [K,C,R,S], physical [K][R][S][C]
        return seg.view(p.shape)

    def materialize(self, with_grads=True):
        self.params = torch.zeros(self.size, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(self.size, dtype=torch.float32, device=self.device) if with_grads else None
        for e in self.entries:
            p = e["param"]
            v = self._view(self.params, e["offset"], p)
            v.copy_(p.data.to(self.device))
            p.data = v
            if with_grads and p.requires_grad:
                p.grad = self._view(self.grads, e["offset"], p)
        return self

    def is_current(self):
        """False if someone re-pointed the parameters (e.g. model.to(...)) since materialize()."""
        if self.params is None:
            return False
        base = self.params.data_ptr()
        for e in self.entries[:3] + self.entries[-3:]:
            if e["param"].data_ptr() != base + 4 * e["offset"]:
                return False
        return True

    def slice(self, off, numel, which="params"):
        return getattr(self, which)[off:off + numel]

    def segments(self, hyper_fn):
        """Groups contiguous entries with equal hyper-parameters: [(offset, numel, hyper)] (padding included)."""
        segs = []
        for i, e in enumerate(self.entries):
            h = hyper_fn(e["name"], e["param"])
            # the end is rounded up to 4 elements only into PADDING: fused heads (pad_after=False) are packed back to back, and a
            # neighbour with other hyper-parameters must not be swept twice (unit_sgd_momentum takes unaligned ranges)
            nxt = self.entries[i + 1]["offset"] if i + 1 < len(self.entries) else self.size
            end = min(_align(e["offset"] + e["numel"], 4), max(nxt, e["offset"] + e["numel"]))
            if segs and segs[-1][2] == h and segs[-1][0] + segs[-1][1] >= e["offset"] - self.ALIGN:
                o, n, _ = segs[-1]
                segs[-1] = (o, end - o, h)
            else:
                segs.append((e["offset"], end - e["offset"], h))
        return segs
