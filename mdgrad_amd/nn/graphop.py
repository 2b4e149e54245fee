"""Pooling / scatter helpers under the reference's names and signatures:
`scatter_add`, `compute_grad` (nff/utils/scatter.py:5-45) and `split_and_sum`, `batch_and_sum` (nff/nn/graphop.py:9-63).

They are what `MessagePassingModule.aggregate` and `SchNet.forward` call here, as in the reference.  The product's
trajectory path does not go through them: with the topology GNNPotentials builds, an interaction block aggregates with the
atom-centric HIP gather (ops.CfconvAggFn, csrc/graph.hip) or the fused block kernels (csrc/cfconv_fused.hip), which need no
scatter and no float atomics."""
import torch


def compute_grad(inputs, output, create_graph=True, retain_graph=True):
    """d(sum of output)/d(inputs), differentiable again by default (nff/utils/scatter.py:5-21)."""
    assert inputs.requires_grad
    g, = torch.autograd.grad(output, inputs, grad_outputs=torch.ones_like(output), create_graph=create_graph,
                             retain_graph=retain_graph)
    return g


def scatter_add(src, index, dim=-1, out=None, dim_size=None, fill_value=0):
    """out[..., index[i], ...] += src[..., i, ...] along `dim` (nff/utils/scatter.py:24-45: same arguments and defaults).
    A 1-D index is broadcast along every other dimension of `src`; without `out` the result has `dim_size` entries along `dim`
    (default: largest index + 1, one host sync) filled with `fill_value` first."""
    dim = dim % src.dim()
    if index.dim() == 1 and src.dim() > 1:
        shape = [1] * src.dim()
        shape[dim] = src.shape[dim]
        index = index.view(shape).expand_as(src)
    if out is None:
        n = int(index.max()) + 1 if dim_size is None else int(dim_size)
        size = list(src.shape)
        size[dim] = n
        out = src.new_full(size, fill_value)
    return out.scatter_add_(dim, index, src)


def split_and_sum(tensor, N):
    """Rows of `tensor` split into consecutive groups of N[0], N[1], ... rows, each summed: [len(N), ...]
    (nff/nn/graphop.py:9-30)."""
    return torch.stack([part.sum(dim=0) for part in torch.split(tensor, list(N))])


def batch_and_sum(dict_input, N, predict_keys, xyz):
    """Per-molecule pooling of the per-atom readout (nff/nn/graphop.py:32-63): every key of `dict_input` that is asked for in
    `predict_keys` -- directly, or through its gradient `key + "_grad"` -- is summed per molecule; `key + "_grad"` is the
    derivative of the pooled value with respect to `xyz` (kept differentiable)."""
    results = {}
    for key, val in dict_input.items():
        want, want_grad = key in predict_keys, (key + "_grad") in predict_keys
        if not (want or want_grad):
            continue
        results[key] = split_and_sum(val, N)
        if want_grad:
            results[key + "_grad"] = compute_grad(inputs=xyz, output=results[key])
    return results
