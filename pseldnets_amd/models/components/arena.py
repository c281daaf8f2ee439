"""Flat parameter arena: every trainable tensor of a network lives in ONE fp32 buffer (plus one gradient buffer,
AdamW moments, and a bf16 shadow when the compute dtype is bf16), laid out in forward order.

Why (MI355X-first): the optimiser, the gradient-norm and the RCCL all-reduce then each touch a single contiguous
range — one kernel / one collective per bucket instead of 188 small ones (the reference steps 188 tensors through
torch.optim.AdamW and clip_grad_norm_) — and backward finishes parameter ranges back to
front, so contiguous tail ranges can be all-reduced while earlier layers are still in backward.
nn.Parameters stay visible under the reference's names: each one is a view into the arena.
"""
import torch

from ... import ops


def _round8(n):
    return (n + 7) // 8 * 8


class ParamArena:
    def __init__(self):
        self.entries = []      # (name, shape, padded_rows_shape)
        self.offsets = {}
        self.size = 0
        self.flat = None       # fp32 master
        self.grad = None
        self.shadow = None     # bf16 copy (bf16 compute mode)
        self.m = self.v = None
        self.step = 0
        self.shadow_valid = False
        self.shadow_t = None   # transposed bf16 copies of the 2-D weights (input-gradient GEMMs)
        self.shadow_t_valid = False
        self._t_desc = None

    def add(self, name, shape, pad_rows=None):
        """Reserve space; `pad_rows` pads dim 0 (zero rows the kernels may read, e.g. the 1530 -> 1536 head)."""
        rows = shape[0] if pad_rows is None else pad_rows
        n = rows
        for d in shape[1:]:
            n *= d
        self.offsets[name] = (self.size, tuple(shape), rows)
        self.entries.append(name)
        self.size += _round8(n)

    def materialize(self, device, init_values):
        self.flat = torch.zeros(self.size, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.size, dtype=torch.float32, device=device)
        for name in self.entries:
            self.view(self.flat, name).copy_(init_values[name])
        self.shadow_valid = False

    def view(self, buf, name, padded=False):
        off, shape, rows = self.offsets[name]
        lead = rows if padded else shape[0]
        n = lead
        for d in shape[1:]:
            n *= d
        return buf[off:off + n].view((lead,) + shape[1:])

    def p(self, name, padded=False):
        return self.view(self.flat, name, padded)

    def g(self, name, padded=False):
        return self.view(self.grad, name, padded)

    def w(self, name, dtype, padded=False):
        """Weight in the compute dtype (bf16 shadow, refreshed lazily, or the fp32 master itself)."""
        if dtype == torch.float32:
            return self.view(self.flat, name, padded)
        if self.shadow is None:
            self.shadow = torch.empty(self.size, dtype=torch.bfloat16, device=self.flat.device)
            self.shadow_valid = False
        if not self.shadow_valid:
            ops.cast_bf16(self.flat, self.shadow)
            self.shadow_valid = True
            self.shadow_t_valid = False
        return self.view(self.shadow, name, padded)

    def wt(self, name, dtype, padded=False):
        """Transposed bf16 copy [cols, rows] of a weight viewed as [rows, prod(rest)] (None in fp32 mode: the fp32 path
        reads the master weights through the transposing loader). All weights are refreshed by ONE batched kernel the
        first time one is asked for after an optimiser step."""
        if dtype != torch.bfloat16:
            return None
        self.w(name, dtype, padded)                       # makes sure the bf16 shadow itself is current
        if self._t_desc is None:
            rows_cols, first = [], 0
            for n in self.entries:
                off, shape, rows = self.offsets[n]
                if len(shape) < 2:
                    continue
                cols = 1
                for d in shape[1:]:
                    cols *= d
                rows_cols += [off, rows, cols, first]
                first += ((rows + 31) // 32) * ((cols + 31) // 32)
            self._t_desc = (torch.tensor(rows_cols, dtype=torch.long, device=self.flat.device), len(rows_cols) // 4, first)
            self.shadow_t = torch.zeros(self.size, dtype=torch.bfloat16, device=self.flat.device)
            self.shadow_t_valid = False
        if not self.shadow_t_valid:
            desc, n_desc, tiles = self._t_desc
            ops.transpose_batch_bf16(self.shadow, self.shadow_t, desc, n_desc, tiles)
            self.shadow_t_valid = True
        off, shape, rows = self.offsets[name]
        lead = rows if padded else shape[0]
        cols = 1
        for d in shape[1:]:
            cols *= d
        # the copy was made with the PADDED row count as the transposed row stride
        return self.shadow_t[off:off + rows * cols].view(cols, rows)[:, :lead] if lead != rows else \
            self.shadow_t[off:off + rows * cols].view(cols, rows)

    def range_of(self, first_name, last_name=None):
        a = self.offsets[first_name][0]
        if last_name is None:
            return a, self.size
        off, shape, rows = self.offsets[last_name]
        n = rows
        for d in shape[1:]:
            n *= d
        return a, off + _round8(n)

    def ensure_opt_state(self):
        if self.m is None:
            self.m = torch.zeros_like(self.flat)
            self.v = torch.zeros_like(self.flat)
