"""Building blocks of the neural-process path with the reference's module interface
(constructor arguments, ``forward`` signatures, ``state_dict`` keys) and HIP execution.

Mirrors, for the hot path only:
  * ``MLP``                     npf/architectures/mlp.py:12-115
  * ``MergeFlatInputs`` / ``merge_flat_input`` (``is_sum_merge=True``)
                                npf/architectures/encoders.py:130-213
  * ``get_attender("scaledot")`` / ``DotAttender``
                                npf/architectures/attention.py:16-86,89-220
``MLP(is_res=True)``, MLP dropout and the concatenating merge are covered too; what else the reference offers
in those files (other activations, attention dropout, the other attention flavours) raises
``NotImplementedError`` here instead of silently running somewhere else.

``forward`` takes row-major device tensors like the reference; internally every module can
also append itself to a :class:`~npf_gwwaveform_amd.chain.Chain` (``append_to``), which is how
the models fuse whole stages into one kernel launch.
"""
from __future__ import annotations

import math
import warnings
from typing import Optional

import torch
import torch.nn as nn

from . import functional as FN
from ._lib import NPF_MAX_FUSED_ROW
from .chain import Chain, PTensor

__all__ = ["MLP", "MergeFlatInputs", "merge_flat_input", "DotAttender", "MultiheadAttender", "TransformerAttender",
           "SelfAttention", "get_attender"]


def _check_relu(activation) -> None:
    if not isinstance(activation, nn.ReLU):
        # the reference itself only works with ReLU-family activations on this path
        # (npf/utils/initialization.py:34-50 raises for anything outside its mapper)
        raise NotImplementedError("the HIP path implements the reference's ReLU MLPs only")


class MLP(nn.Module):
    """``MLP(input_size, output_size, hidden_size=32, n_hidden_layers=1, ...)``: to_hidden ->
    ReLU -> (n_hidden_layers - 1) x [Linear -> ReLU] -> out; hidden size clamped as in
    npf/architectures/mlp.py:64-79."""

    def __init__(self, input_size, output_size, hidden_size=32, n_hidden_layers=1, activation=None, is_bias=True,
                 dropout=0, is_force_hid_smaller=False, is_res=False):
        super().__init__()
        activation = nn.ReLU() if activation is None else activation
        _check_relu(activation)
        if not 0 <= dropout < 1:
            raise ValueError(f"dropout probability has to be in [0, 1), got {dropout}")
        self.input_size, self.output_size, self.n_hidden_layers, self.is_res = input_size, output_size, n_hidden_layers, is_res
        self.hidden_size = hidden_size
        if is_force_hid_smaller and self.hidden_size > max(output_size, input_size):
            self.hidden_size = max(output_size, input_size)
            warnings.warn(f"hidden_size={hidden_size} larger than output={output_size} and input={input_size}. "
                          f"Setting it to {self.hidden_size}.")
        elif self.hidden_size < min(output_size, input_size):
            self.hidden_size = min(output_size, input_size)
            warnings.warn(f"hidden_size={hidden_size} smaller than output={output_size} and input={input_size}. "
                          f"Setting it to {self.hidden_size}.")
        self.dropout_p = float(dropout)
        self.dropout = nn.Dropout(p=dropout) if dropout > 0 else nn.Identity()  # (interface; the mask is applied in-kernel)
        self.activation = activation
        self.to_hidden = nn.Linear(input_size, self.hidden_size, bias=is_bias)
        self.linears = nn.ModuleList(
            [nn.Linear(self.hidden_size, self.hidden_size, bias=is_bias) for _ in range(n_hidden_layers - 1)])
        self.out = nn.Linear(self.hidden_size, output_size, bias=is_bias)
        self.reset_parameters()

    def reset_parameters(self):
        """Effective initialisation of the reference (SURVEY.md 8a row 12;
        npf/utils/initialization.py:34-94): hidden layers keep torch's default weight
        init with zeroed biases, ``out`` gets kaiming-uniform(relu) and a zero bias."""
        for lin in [self.to_hidden, *self.linears]:
            if lin.bias is not None:
                lin.bias.data.zero_()
        if self.out.bias is not None:
            self.out.bias.data.zero_()
        nn.init.kaiming_uniform_(self.out.weight, nonlinearity="relu")

    def layers(self):
        return [self.to_hidden, *self.linears, self.out]

    def append_to(self, ch: Chain, skip_first: bool = False, skip_last: bool = False) -> Chain:
        """cur <- MLP(cur).  ``skip_first``: cur already holds the activated output of ``to_hidden`` (a
        concatenating merge computes that layer as two accumulating halves); ``skip_last``: stop in front of
        ``out``.  With ``is_res`` every hidden-to-hidden layer adds its input back after the activation
        (mlp.py:100-104)."""
        ls = self.layers()
        for j, lin in enumerate(ls):
            if j == len(ls) - 1 and skip_last:
                continue
            if j == 0 and skip_first:
                # (the caller ran ``to_hidden`` + ReLU itself; the dropout behind it, mlp.py:96-98, is still this module's)
                if len(ls) > 1 and self.dropout_p > 0 and self.training:
                    ch.dropout(self._sign_mask(ch, lin.out_features), self.dropout_p)
                continue
            W = lin.weight
            if j == 0 and W.shape[1] % 4 != 0 and W.shape[1] == ch.F:
                # skinny first layers (x: 1-2 features, y: 2): zero-pad the fan-in to a multiple of 4
                # so that the weight rows are 16-byte aligned and stream through the 16-byte LDS-DMA
                # path (the padded input features are zero, so the result is unchanged)
                pad = -W.shape[1] % 4
                W = torch.nn.functional.pad(W, (0, pad))
                ch.F = W.shape[1]
            ch.linear(W, lin.bias, relu=(j < len(ls) - 1), residual=(self.is_res and 0 < j < len(ls) - 1))
            if j < len(ls) - 1 and self.dropout_p > 0 and self.training:
                ch.dropout(self._sign_mask(ch, lin.out_features), self.dropout_p)  # mlp.py:98,105: after activation (+ residual)
        return ch

    # tests: an iterator of row-major keep masks [n_tasks * pts, F] (or [n_tasks, pts, F]) consumed in call order instead
    # of the device generator (the reference draws its masks from torch's CPU generator, which no GPU draw reproduces)
    mask_source = None

    def _sign_mask(self, ch: Chain, F: int) -> torch.Tensor:
        """+1 (keep, probability 1 - p) / -1 (drop) per (task, point, unit) as a PT32 tensor."""
        from .chain import pt_shape

        if MLP.mask_source is not None:
            keep = next(MLP.mask_source).to(ch.device).reshape(ch.n_tasks, ch.pts, F).float()
            return FN.pack_pt(keep * 2.0 - 1.0)
        keep = torch.rand(pt_shape(ch.n_tasks, ch.pts, F), device=ch.device) < (1.0 - self.dropout_p)
        return torch.where(keep, 1.0, -1.0)

    def forward(self, x):
        lead, n_in = x.shape[:-1], x.shape[-1]
        rows = int(math.prod(lead)) if len(lead) else 1
        if rows == 0:
            return x.new_zeros(*lead, self.output_size)
        ch = Chain(1, rows, x.device)
        ch.input_pt(FN.pack_pt(x.reshape(1, rows, n_in)), n_in)
        if self.output_size <= NPF_MAX_FUSED_ROW:
            self.append_to(ch).output_pt()
            (y,) = ch.run()
            return FN.unpack_pt(y, rows, self.output_size).reshape(*lead, self.output_size)
        # wide output layer (e.g. the latent encoder's r -> 2 z): run the trunk once, then the
        # output layer in row blocks of <= 256 outputs
        (h,) = self.append_to(ch, skip_last=True).output_pt().run()
        outs = []
        for lo in range(0, self.output_size, NPF_MAX_FUSED_ROW):
            hi = min(lo + NPF_MAX_FUSED_ROW, self.output_size)
            c2 = Chain(1, rows, x.device)
            c2.input_pt(h, self.hidden_size)
            c2.linear(self.out.weight[lo:hi], self.out.bias[lo:hi] if self.out.bias is not None else None).output_pt()
            outs.append(FN.unpack_pt(c2.run()[0], rows, hi - lo))
        return torch.cat(outs, dim=-1).reshape(*lead, self.output_size)


class MergeFlatInputs(nn.Module):
    """Two-input wrapper (npf/architectures/encoders.py:130-183).  Sum merge (``is_sum_merge=True``, what
    the reference's encoders / decoders use): ``flat_module(relu(x1 + resizer(x2)))``.  Concatenating merge
    (the constructor default): ``flat_module(cat(x1, x2))`` -- the concatenation is never materialised: the
    first layer of the flat module runs as two accumulating halves, ``W[:, :x1_dim] x1 + W[:, x1_dim:] x2 + b``."""

    def __init__(self, FlatModule, x1_dim, x2_dim, n_out, is_sum_merge=False, **kwargs):
        super().__init__()
        self.is_sum_merge = bool(is_sum_merge)
        self.x1_dim, self.x2_dim = x1_dim, x2_dim
        if self.is_sum_merge:
            self.resizer = MLP(x2_dim, x1_dim)
            self.flat_module = FlatModule(x1_dim, n_out, **kwargs)
        else:
            self.flat_module = FlatModule(x1_dim + x2_dim, n_out, **kwargs)
        if not isinstance(self.flat_module, (MLP, SelfAttention)):
            raise NotImplementedError("the HIP path needs an MLP or a SelfAttention as the flat module")
        if not self.is_sum_merge and not isinstance(self.flat_module, MLP):
            raise NotImplementedError("a concatenating merge needs an MLP flat module on the HIP path")

    def reset_parameters(self):  # the reference's weights_init is a no-op here
        pass

    def _halves(self):
        """(W1, W2, b) of the concatenating merge's first layer: column slices of ``to_hidden.weight``."""
        lin = self.flat_module.to_hidden
        return lin.weight[:, : self.x1_dim], lin.weight[:, self.x1_dim:], lin.bias

    def _x1_half(self, x1_pt: torch.Tensor, pts: int) -> torch.Tensor:
        """W1 x1 of a concatenating merge as a PT32 tensor (its own launch: x1 is not the chain's cur)."""
        ch = Chain(x1_pt.shape[0], pts, x1_pt.device)
        ch.input_pt(x1_pt, self.x1_dim).linear(self._halves()[0], None).output_pt()
        return ch.run()[0]

    def append_to(self, ch: Chain, x1_pt: Optional[torch.Tensor] = None, x1_modulus: int = 0,
                  x1_taskvec: bool = False, x1_rm: bool = False) -> Chain:
        """cur = x2 on entry.  ``x1_pt``: PT32 tensor added before the ReLU (``x1_rm``: it is a
        row-major [tasks, pts, x1_dim] tensor instead)."""
        if isinstance(self.flat_module, SelfAttention):
            raise NotImplementedError("a SelfAttention flat module is not one chain: use run_pt")
        if not self.is_sum_merge:
            if x1_rm:
                raise NotImplementedError("row-major x1 with a concatenating merge")
            _, W2, b = self._halves()
            ch.linear(W2, b, relu=True, addend=self._x1_half(x1_pt, ch.pts), addend_modulus=x1_modulus)
            return self.flat_module.append_to(ch, skip_first=True)
        rl = self.resizer.layers()
        for j, lin in enumerate(rl[:-1]):
            ch.linear(lin.weight, lin.bias, relu=True)
        ch.linear(rl[-1].weight, rl[-1].bias, relu=True, addend=x1_pt, addend_modulus=x1_modulus, addend_rm=x1_rm)
        return self.flat_module.append_to(ch)

    def append_taskvec_to(self, ch: Chain, vec_rows: torch.Tensor, modulus: int = 0) -> Chain:
        """cur = x1 on entry; x2 is one row-major vector per task, ``vec_rows`` [n_tasks, x2_dim] (np.py:107-110,161:
        the reference expands it over the targets and recomputes its half per target; here that half runs
        once per task and enters as a per-task vector)."""
        if not isinstance(self.flat_module, MLP):
            raise NotImplementedError("a per-task x2 needs an MLP flat module")
        if self.is_sum_merge:
            tv = self.resizer(vec_rows)
        else:
            W1, W2, b = self._halves()
            tv = _rows_linear(W2, b, vec_rows)
            ch.linear(W1, None)
        Fp = -(-tv.shape[1] // 32) * 32
        if tv.shape[1] != Fp:
            tv = torch.nn.functional.pad(tv, (0, Fp - tv.shape[1]))
        ch.add_taskvec(tv.contiguous(), relu=True, modulus=modulus)
        return self.flat_module.append_to(ch, skip_first=not self.is_sum_merge)

    # ---- the flat MLP's 256 -> 256 layers on the split kernel (mlp_x6.py): fp32 results on the bf16 matrix pipe ----
    def _x6_stack(self):
        """(layers, relus, covers_out) of the flat module that can run on ``npf_mlp_x6_run``, or None: a sum-merge MLP
        whose input and hidden width are 256, no residual, no active dropout.  ``covers_out``: ``out`` is 256 -> 256 too."""
        from . import mlp_x6

        fm = self.flat_module
        if not (self.is_sum_merge and isinstance(fm, MLP)) or fm.is_res or (fm.dropout_p > 0 and fm.training):
            return None
        stack = [fm.to_hidden, *fm.linears]
        if not mlp_x6.usable(stack):
            return None
        relus = [True] * len(stack)
        covers_out = mlp_x6.usable([fm.out])
        if covers_out:
            stack, relus = stack + [fm.out], relus + [False]
        return stack, relus, covers_out

    def x6_resizer_ok(self, x1_modulus: int = 0) -> bool:
        """Will ``finish_rows`` put the resizer and the merge on the split kernel too (x1 as the addend of its last layer)."""
        from . import mlp_x6

        st = self._x6_stack()
        return st is not None and not st[2] and x1_modulus == 0 and mlp_x6.usable(self.resizer.layers())

    def _merge_only(self, ch: Chain, x1_pt=None, x1_modulus: int = 0, taskvec=None) -> Chain:
        """cur <- relu(x1 + resizer(x2)): ``x1_pt`` with cur = x2, or ``taskvec`` (x2 one vector per task) with cur = x1."""
        if taskvec is not None:
            tv = self.resizer(taskvec)
            Fp = -(-tv.shape[1] // 32) * 32
            if tv.shape[1] != Fp:
                tv = torch.nn.functional.pad(tv, (0, Fp - tv.shape[1]))
            return ch.add_taskvec(tv.contiguous(), relu=True)
        rl = self.resizer.layers()
        for lin in rl[:-1]:
            ch.linear(lin.weight, lin.bias, relu=True)
        return ch.linear(rl[-1].weight, rl[-1].bias, relu=True, addend=x1_pt, addend_modulus=x1_modulus)

    def finish_rows(self, ch: Chain, x1_pt=None, x1_modulus: int = 0, taskvec=None) -> torch.Tensor:
        """Finish ``ch`` with this module and run it: the row-major [n_tasks, pts, n_out] output (n_out <= 32: the
        decoder's sufficient statistics)."""
        from . import mlp_x6

        st = self._x6_stack()
        if st is None or st[2]:
            if taskvec is not None:
                return self.append_taskvec_to(ch, taskvec).output_rows().run()[0]
            return self.append_to(ch, x1_pt=x1_pt, x1_modulus=x1_modulus).output_rows().run()[0]
        stack, relus, _ = st
        rl = self.resizer.layers()
        out = self.flat_module.out
        if x1_pt is not None and x1_modulus == 0 and taskvec is None and mlp_x6.usable(rl):
            # the resizer and the merge on the split kernel as well: x1 enters the resizer's last layer as its addend
            outs = ch.output_pt().run()
            x2 = outs[-1]
            if getattr(ch, "x1_tapped", False):  # (x1 came through the chain: its gradient goes back into that chain's dgrad)
                x1_pt = outs[0]
            tail = out if mlp_x6.tail_usable(out, ch.pts) else None
            h = mlp_x6.run_stack(x2, ch.pts, rl + stack, [True] * len(rl) + relus, addend=x1_pt, add_at=len(rl) - 1, tail=tail)
        else:
            (h0,) = self._merge_only(ch, x1_pt, x1_modulus, taskvec).output_pt().run()
            tail = out if mlp_x6.tail_usable(out, ch.pts) else None
            h = mlp_x6.run_stack(h0, ch.pts, stack, relus, tail=tail)
        if tail is not None:
            return h  # (the rows already)
        ch2 = Chain(ch.n_tasks, ch.pts, ch.device)
        ch2.input_pt(h, out.in_features).linear(out.weight, out.bias).output_rows()
        return ch2.run()[0]

    def run_pt(self, ch: Chain, x1_pt, n_tasks: int, pts: int, with_tr: bool = False, **kw) -> PTensor:
        """Finish ``ch`` (cur = x2) with this module and run it: a :class:`PTensor` [n_tasks, pts, n_out]
        (carrying the feature-major copy / bf16 images when ``with_tr``).  One launch for an MLP flat
        module; for a SelfAttention flat module the merge is a launch and the attention layers follow."""
        if isinstance(self.flat_module, MLP):
            st = self._x6_stack()
            if st is not None and set(kw) <= {"x1_modulus"}:
                from . import mlp_x6

                stack, relus, covers_out = st
                (h0,) = self._merge_only(ch, x1_pt, kw.get("x1_modulus", 0)).output_pt().run()
                h = mlp_x6.run_stack(h0, pts, stack, relus)
                out = self.flat_module.out
                if covers_out:
                    if not with_tr:
                        return PTensor(h, pts, out.out_features)
                    from . import chain as _chain

                    if _chain.COMPUTE_DTYPE == "fp32":
                        # only the feature-major copy is missing: a chain that reads h and stores it transposed (no PT32
                        # output of its own: h itself is the tensor, and its gradient goes straight to the stack)
                        ch2 = Chain(n_tasks, pts, ch.device)
                        ch2.input_pt(h.detach(), out.out_features).store_tr()
                        (tr,) = ch2.run()
                        return PTensor(h, pts, out.out_features, tr=tr)
                ch2 = Chain(n_tasks, pts, ch.device)
                ch2.input_pt(h, out.in_features)
                if not covers_out:
                    ch2.linear(out.weight, out.bias)
                return ch2.run_pt(as_weights=with_tr)
            return self.append_to(ch, x1_pt=x1_pt, **kw).run_pt(as_weights=with_tr)
        rl = self.resizer.layers()
        for lin in rl[:-1]:
            ch.linear(lin.weight, lin.bias, relu=True)
        ch.linear(rl[-1].weight, rl[-1].bias, relu=True, addend=x1_pt, addend_modulus=kw.get("x1_modulus", 0)).output_pt()
        return self.flat_module.forward_pt(ch.run()[0], n_tasks, pts, with_tr=with_tr)

    def forward(self, x1, x2):
        # row-major API: x1 [..., T, x1_dim]; x2 broadcastable to it with optional extra leading dims
        if x2.dim() < x1.dim():
            raise NotImplementedError("x2 must have at least as many dims as x1")
        T, d1 = x1.shape[-2], x1.shape[-1]
        lead2 = x2.shape[:-2]
        n2 = int(math.prod(lead2)) if len(lead2) else 1
        n1 = int(math.prod(x1.shape[:-2])) if x1.dim() > 2 else 1
        if n2 % max(n1, 1) != 0:
            raise NotImplementedError("unsupported broadcast between x1 and x2")
        x2 = x2.expand(*lead2, T, x2.shape[-1]) if x2.shape[-2] != T else x2
        n_out = self.flat_module.out_dim if isinstance(self.flat_module, SelfAttention) else self.flat_module.output_size
        if T == 0 or n2 == 0:
            return x1.new_zeros(*lead2, T, n_out)
        if isinstance(self.flat_module, SelfAttention):
            ch = Chain(n2, T, x1.device)
            ch.input_pt(FN.pack_pt(x2.reshape(n2, T, x2.shape[-1])), x2.shape[-1])
            y = self.run_pt(ch, FN.pack_pt(x1.reshape(n1, T, d1)), n2, T, x1_modulus=(n1 if n1 != n2 else 0))
            return FN.unpack_pt(y.t, T, n_out).reshape(*lead2, T, n_out)
        ch = Chain(n2, T, x1.device)
        d2 = x2.shape[-1]
        no_grad = not torch.is_grad_enabled() or not (x1.requires_grad or x2.requires_grad or
                                                      any(p.requires_grad for p in self.parameters()))
        if self.is_sum_merge and no_grad and n1 == n2 and n_out <= 4:
            from . import x6

            x1r, x2r = x1.reshape(n1, T, d1), x2.reshape(n2, T, d2)
            if x6.decode_rows_usable(self, x1r, x2r):
                # inference at a width the x6 programs cover (128 / 256 / 512): merge, decoder and output layer as ONE launch on
                # the bf16 matrix pipe (fp32 results), straight from the row-major tensors
                return x6.decode_rows(self, x1r, x2r).reshape(*lead2, T, n_out)
        if self.is_sum_merge and no_grad and d1 % 32 == 0 and d2 % 32 == 0:
            # inference: the row-major module-boundary tensors go straight into the chain (no PT32
            # packing pass over x1 and x2)
            ch.input_rm(x2.reshape(n2, T, d2).contiguous(), d2)
            self.append_to(ch, x1.reshape(n1, T, d1).contiguous(), x1_modulus=(n1 if n1 != n2 else 0), x1_rm=True)
        else:
            ch.input_pt(FN.pack_pt(x2.reshape(n2, T, d2)), d2)
            self.append_to(ch, FN.pack_pt(x1.reshape(n1, T, d1)), x1_modulus=(n1 if n1 != n2 else 0))
        if n_out <= 32:
            (y,) = ch.output_rows().run()
            return y[..., :n_out].reshape(*lead2, T, n_out)
        (y,) = ch.output_pt().run()
        return FN.unpack_pt(y, T, n_out).reshape(*lead2, T, n_out)


def _rows_linear(W: torch.Tensor, b: Optional[torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """x [rows, K] -> x W^T + b [rows, N] as one chain launch over the rows."""
    rows, K = x.shape
    ch = Chain(1, rows, x.device)
    ch.input_pt(FN.pack_pt(x.reshape(1, rows, K)), K).linear(W, b).output_pt()
    return FN.unpack_pt(ch.run()[0], rows, W.shape[0]).reshape(rows, W.shape[0])


def merge_flat_input(module, is_sum_merge=False, **kwargs):
    """Factory with the reference's calling convention
    (npf/architectures/encoders.py:186-213): ``merge_flat_input(MLP, is_sum_merge=True)(x_dim, flat_dim, n_out)``."""

    def merged_flat_input(x_shape, flat_dim, n_out, **kwargs2):
        assert isinstance(x_shape, int)
        return MergeFlatInputs(module, x_shape, flat_dim, n_out, is_sum_merge=is_sum_merge, **kwargs2, **kwargs)

    return merged_flat_input


class DotAttender(nn.Module):
    """Scaled dot-product cross attention without learned projections:
    ``softmax(Q K^T / sqrt(d)) V`` (npf/architectures/attention.py:89-220)."""

    def __init__(self, kq_size, value_size, out_size, is_scale=True, is_normalize=True, dropout=0):
        super().__init__()
        if not is_normalize or dropout != 0:
            raise NotImplementedError("un-normalised or dropout attention is not on the hot path")
        if value_size != out_size:
            raise NotImplementedError("attention output resizer (value_size != out_size) is not on the hot path")
        self.kq_size, self.value_size, self.out_size, self.is_scale = kq_size, value_size, out_size, is_scale
        self.is_normalize, self.is_resize = True, False
        self.dropout = nn.Identity()

    def append_to(self, ch: Chain, keys_pt, values_pt, n_keys: int, keys_tr=None, values_tr=None, keys_img=None,
                  values_img=None) -> Chain:
        """cur = queries on entry, context vectors on exit.  ``keys_tr`` / ``values_tr``:
        feature-major copies of the keys / values (``Chain.store_tr``) for the DMA fast path;
        ``keys_img`` / ``values_img``: their bf16 images (``Chain.store_bf16_images``, bf16 mode)."""
        scale = 1.0 / math.sqrt(self.kq_size) if self.is_scale else 1.0
        return (ch.attn_scores(keys_pt, n_keys, keys_tr=keys_tr, keys_img=keys_img).softmax(scale)
                .attn_values(values_pt, self.value_size, values_tr=values_tr, values_img=values_img))

    def fits_fused(self, n_keys: int) -> bool:
        """Can ``append_to`` keep a whole score row in registers (else: ``attend_pt``)."""
        return n_keys <= NPF_MAX_FUSED_ROW

    def attend_pt(self, queries_pt, keys_pt, values_pt, n_keys: int, n_queries: int, keys_tr=None, values_tr=None):
        """PT32 in, PT32 out, any number of keys (fused chain up to 256 keys, blocked softmax of
        attention_long.py beyond)."""
        from .attention_long import long_scaledot_attention

        if self.fits_fused(n_keys):
            ch = Chain(queries_pt.shape[0], n_queries, queries_pt.device, wg_per_task=True)
            ch.input_pt(queries_pt, self.kq_size)
            self.append_to(ch, keys_pt, values_pt, n_keys, keys_tr=keys_tr, values_tr=values_tr).output_pt()
            return ch.run()[0]

        scale = 1.0 / math.sqrt(self.kq_size) if self.is_scale else 1.0
        return long_scaledot_attention(queries_pt, keys_pt, values_pt, n_keys, n_queries, self.value_size, scale,
                                       k_tr=keys_tr, v_tr=values_tr, d=self.kq_size)

    def forward(self, keys, queries, values):
        B, C, d = keys.shape
        T = queries.shape[1]
        if keys.dim() != 3 or queries.dim() != 3:
            raise NotImplementedError("relative-position (4-D) keys are not on the hot path")
        if C == 0:
            raise ValueError("attention over zero keys")
        if not self.fits_fused(C):
            o = self.attend_pt(FN.pack_pt(queries), FN.pack_pt(keys), FN.pack_pt(values), C, T)
            return FN.unpack_pt(o, T, self.out_size)
        ch = Chain(B, T, keys.device, wg_per_task=True)
        ch.input_pt(FN.pack_pt(queries), d)
        self.append_to(ch, FN.pack_pt(keys), FN.pack_pt(values), C).output_pt()
        (o,) = ch.run()
        return FN.unpack_pt(o, T, self.out_size)


class MultiheadAttender(nn.Module):
    """Multihead attention (npf/architectures/attention.py:375-527): K (no bias) / Q (bias) / V (no
    bias) projections, ``n_heads`` scaled-dot attentions on the head slices (heads stacked as extra
    tasks, scale 1/sqrt(head size)), concatenation, optional post Linear.  Same parameter names as
    the reference: ``key_transform``, ``query_transform``, ``value_transform``, ``post_processor``."""

    def __init__(self, kq_size, value_size, out_size, n_heads=8, is_post_process=True, dropout=0, is_relative_pos=False):
        super().__init__()
        if dropout != 0 or is_relative_pos:
            raise NotImplementedError("attention dropout / relative positions are not on the hot path")
        if kq_size != value_size:
            raise NotImplementedError("the HIP path needs kq_size == value_size (always the case in AttnCNP / AttnLNP)")
        assert kq_size % n_heads == 0, "{} % {} != 0".format(kq_size, n_heads)
        assert value_size % n_heads == 0, "{} % {} != 0".format(value_size, n_heads)
        if (kq_size // n_heads) % 4 != 0:
            raise NotImplementedError("head size must be a multiple of 4 on the HIP path")
        self.is_relative_pos = False
        self.key_transform = nn.Linear(kq_size, kq_size, bias=False)
        self.query_transform = nn.Linear(kq_size, kq_size, bias=True)
        self.value_transform = nn.Linear(value_size, value_size, bias=False)
        self.n_heads = n_heads
        self.kq_head_size = kq_size // n_heads
        self.value_head_size = kq_size // n_heads
        self.kq_size, self.value_size, self.out_size = kq_size, value_size, out_size
        self.dot = DotAttender(self.kq_head_size, self.value_head_size, self.value_head_size, is_scale=True)
        self.post_processor = nn.Linear(value_size, out_size) if is_post_process or value_size != out_size else None
        self.reset_parameters()

    def reset_parameters(self):
        # attention.py:444-455 (weights_init itself is a no-op)
        std = math.sqrt(2.0 / (self.kq_size + self.kq_head_size))
        nn.init.normal_(self.key_transform.weight, mean=0, std=std)
        nn.init.normal_(self.query_transform.weight, mean=0, std=std)
        std = math.sqrt(2.0 / (self.value_size + self.value_head_size))
        nn.init.normal_(self.value_transform.weight, mean=0, std=std)

    # -- PT32 level
    def _project(self, x_pt, n_tasks, pts, lin):
        ch = Chain(n_tasks, pts, x_pt.device)
        ch.input_pt(x_pt, lin.in_features).linear(lin.weight, lin.bias).output_pt()
        return ch.run()[0]

    def _heads_attention(self, queries_pt, keys_pt, values_pt, B, C, T, queries_proj=None):
        """K/Q/V projections, per-head scaled-dot attention, heads merged: PT32 [B, T, value_size].  ``queries_proj``: the query
        projection when the launch that encoded the queries already made it (x6.xenc_proj)."""
        H, d = self.n_heads, self.kq_size
        if FN.mha_usable(self.kq_head_size, self.value_head_size, C):
            # 16- / 32-feature heads (the reference's default r_dim = 128 with 8 heads; 256 with 8): one launch on the projected
            # tensors, the heads are feature slices of the PT32 tiles (csrc/mha_kernel.hip)
            from . import x6

            if x6.pair_linear_usable(self.key_transform, self.value_transform):  # (both on the context points: one launch)
                Kp, Vp = x6.pair_linear(keys_pt, values_pt, C, self.key_transform, self.value_transform)
            else:
                Kp, Vp = self._project(keys_pt, B, C, self.key_transform), self._project(values_pt, B, C, self.value_transform)
            Qp = queries_proj if queries_proj is not None else self._project(queries_pt, B, T, self.query_transform)
            return FN.mha(Qp, Kp, Vp, B, C, T, H, self.kq_head_size)
        Kh = FN.split_heads(self._project(keys_pt, B, C, self.key_transform), B, C, d, H)
        Qh = FN.split_heads(self._project(queries_pt, B, T, self.query_transform), B, T, d, H)
        Vh = FN.split_heads(self._project(values_pt, B, C, self.value_transform), B, C, self.value_size, H)
        if self.dot.fits_fused(C):
            ch = Chain(H * B, T, queries_pt.device, wg_per_task=True)
            ch.input_pt(Qh, self.kq_head_size)
            self.dot.append_to(ch, Kh, Vh, C).output_pt()
            Oh = ch.run()[0]
        else:
            Oh = self.dot.attend_pt(Qh, Kh, Vh, C, T)
        return FN.merge_heads(Oh, B, T, self.value_size, H)

    def attend_pt(self, queries_pt, keys_pt, values_pt, n_keys: int, n_queries: int, keys_tr=None, values_tr=None,
                  queries_proj=None):
        B = queries_pt.shape[0]
        ctx = self._heads_attention(queries_pt, keys_pt, values_pt, B, n_keys, n_queries, queries_proj)
        if self.post_processor is not None:
            ctx = self._project(ctx, B, n_queries, self.post_processor)
        return ctx

    def forward(self, keys, queries, values, rel_pos_enc=None, **kwargs):
        if rel_pos_enc is not None or keys.dim() != 3:
            raise NotImplementedError("relative position encodings are not on the hot path")
        C, T = keys.shape[1], queries.shape[1]
        if C == 0:
            raise ValueError("attention over zero keys")
        o = self.attend_pt(FN.pack_pt(queries), FN.pack_pt(keys), FN.pack_pt(values), C, T)
        return FN.unpack_pt(o, T, self.out_size)


class TransformerAttender(MultiheadAttender):
    """Image-transformer style attention (attention.py:530-588): multihead attention without the
    post Linear, ``LayerNorm(context + queries)``, then ``LayerNorm(x + MLP(x))``.  This is what the
    reference's notebooks and every shipped ``Attn*`` checkpoint use."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, is_post_process=False, **kwargs)
        assert self.kq_size == self.out_size
        self.layer_norm1 = nn.LayerNorm(self.out_size)
        self.layer_norm2 = nn.LayerNorm(self.out_size)
        self.mlp = MLP(self.out_size, self.out_size, hidden_size=self.out_size, activation=nn.ReLU())
        self.reset_parameters()

    def attend_pt(self, queries_pt, keys_pt, values_pt, n_keys: int, n_queries: int, keys_tr=None, values_tr=None,
                  queries_proj=None):
        B, T, d = queries_pt.shape[0], n_queries, self.out_size
        ctx = self._heads_attention(queries_pt, keys_pt, values_pt, B, n_keys, T, queries_proj)
        ln1, ln2 = self.layer_norm1, self.layer_norm2
        if FN.add_layernorm_usable(d):
            x = FN.add_layernorm(ctx, queries_pt, ln1, B, T)  # (a bandwidth-bound kernel of its own: csrc/ln_kernel.hip)
        else:
            ch = Chain(B, T, queries_pt.device)
            ch.input_pt(ctx, d).add_pt(queries_pt).layernorm(ln1.weight, ln1.bias, ln1.eps).output_pt()
            x = ch.run()[0]
        from . import x6

        if FN.add_layernorm_usable(d) and x6.mlp_pt_usable(self.mlp):
            # the MLP block as one x6 program each way, LayerNorm(x + MLP(x)) on the bandwidth-bound kernel
            return FN.add_layernorm(x, x6.mlp_pt(self.mlp, x, T), ln2, B, T)
        # (the residual re-reads x, so the MLP block is its own launch)
        ls = self.mlp.layers()
        ch = Chain(B, T, queries_pt.device)
        ch.input_pt(x, d)
        for lin in ls[:-1]:
            ch.linear(lin.weight, lin.bias, relu=True)
        ch.linear(ls[-1].weight, ls[-1].bias, addend=x).layernorm(ln2.weight, ln2.bias, ln2.eps).output_pt()
        return ch.run()[0]


class SelfAttention(nn.Module):
    """Stack of self-attention layers over the points of each task (npf/architectures/selfattn.py:10-100):
    ``out = layer(out, out, out)`` for ``n_attn_layers`` attenders, then an optional resize Linear.
    Positional encodings (absolute / relative) are not on the hot path."""

    def __init__(self, x_dim, out_dim=None, n_attn_layers=2, attention="transformer", positional=None, position_dim=None,
                 max_len=2000, **kwargs):
        super().__init__()
        if positional is not None:
            raise NotImplementedError("positional encodings are not on the hot path")
        self.positional = None
        self.attn_layers = nn.ModuleList([get_attender(attention, x_dim, x_dim, x_dim, **kwargs) for _ in range(n_attn_layers)])
        for layer in self.attn_layers:
            if not hasattr(layer, "attend_pt"):
                raise NotImplementedError("self attention needs one of the attenders of this package")
        self.is_resize = out_dim is not None
        self.x_dim, self.out_dim = x_dim, (out_dim if out_dim is not None else x_dim)
        if self.is_resize:
            self.resize = nn.Linear(x_dim, out_dim)

    def reset_parameters(self):  # weights_init is a no-op
        pass

    def forward_pt(self, x_pt, n_tasks: int, pts: int, with_tr: bool = False) -> PTensor:
        out = x_pt
        for layer in self.attn_layers:
            out = layer.attend_pt(out, out, out, pts, pts)
        if self.is_resize or with_tr:
            ch = Chain(n_tasks, pts, x_pt.device)
            ch.input_pt(out, self.x_dim)
            if self.is_resize:
                ch.linear(self.resize.weight, self.resize.bias)
            return ch.run_pt(as_weights=with_tr)
        return PTensor(out, pts, self.x_dim)

    def forward(self, X, positions=None):
        if positions is not None:
            raise NotImplementedError("positional encodings are not on the hot path")
        lead, P, d = X.shape[:-2], X.shape[-2], X.shape[-1]
        n = int(math.prod(lead)) if len(lead) else 1
        out = self.forward_pt(FN.pack_pt(X.reshape(n, P, d)), n, P)
        return FN.unpack_pt(out.t, P, self.out_dim).reshape(*lead, P, self.out_dim)


def get_attender(attention, kq_size, value_size, out_size, **kwargs):
    """``get_attender`` of npf/architectures/attention.py:16-86 for the hot path."""
    if not isinstance(attention, str):
        return attention(kq_size, value_size, out_size, **kwargs)
    attention = attention.lower()
    if attention == "scaledot":
        return DotAttender(kq_size, value_size, out_size, is_scale=True, **kwargs)
    if attention == "multihead":
        return MultiheadAttender(kq_size, value_size, out_size, **kwargs)
    if attention == "transformer":
        return TransformerAttender(kq_size, value_size, out_size, **kwargs)
    if attention in ("multiplicative", "additive", "cosine", "manhattan", "euclidean", "weighted_dist"):
        raise NotImplementedError(f"attention={attention!r} is not on the MI355X hot path "
                                  "(scaledot, multihead, transformer are)")
    raise ValueError("Unknown attention method {}".format(attention))
