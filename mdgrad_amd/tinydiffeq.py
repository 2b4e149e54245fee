"""Fixed-grid ODE machinery (the trimmed torchdiffeq the reference carries in
torchmd/tinydiffeq.py): solver base class, RK4 (pure torch; used by out-of-scope demos), the
flatten helpers and the input check that turns a decreasing time grid into a reversed
function (tinydiffeq.py:121-143)."""
import torch


def _flatten(sequence):                                      # tinydiffeq.py:106-108
    flat = [p.contiguous().view(-1) for p in sequence]
    return torch.cat(flat) if len(flat) > 0 else torch.tensor([])


def _flatten_convert_none_to_zeros(sequence, like_sequence):  # tinydiffeq.py:111-116
    flat = [p.contiguous().view(-1) if p is not None else torch.zeros_like(q).view(-1)
            for p, q in zip(sequence, like_sequence)]
    return torch.cat(flat) if len(flat) > 0 else torch.tensor([])


def _check_inputs(func, y0, t):
    tensor_input = torch.is_tensor(y0)
    if tensor_input:
        y0 = (y0,)
        inner = func
        func = lambda t, y: (inner(t, y[0]),)
    if not isinstance(y0, tuple):
        raise AssertionError('y0 must be either a torch.Tensor or a tuple')
    for y in y0:
        if not torch.is_tensor(y):
            raise AssertionError('each element must be a torch.Tensor but received {}'.format(type(y)))
        if not torch.is_floating_point(y):
            raise TypeError('`y0` must be a floating point Tensor but is a {}'.format(y.type()))
    if not torch.is_floating_point(t):
        raise TypeError('`t` must be a floating point Tensor but is a {}'.format(t.type()))
    if bool((t[1:] < t[:-1]).all()) and t.numel() > 1:
        # integrate backwards in time: s = -t, dy/ds = -f(-s, y)          tinydiffeq.py:132-135
        fwd = func
        t = -t
        func = lambda s, y: tuple(-f for f in fwd(-s, y))
    return tensor_input, func, y0, t


class FixedGridODESolver:
    """Explicit fixed-step solver on the user's time grid (tinydiffeq.py:13-85).  Subclasses
    provide step_func(func, t, dt, y) -> tuple of increments."""

    def __init__(self, func, y0, step_size=None, grid_constructor=None, **unused):
        unused.pop('rtol', None)
        unused.pop('atol', None)
        if unused:
            import warnings
            warnings.warn('{}: Unexpected arguments {}'.format(self.__class__.__name__, unused))
        if step_size is not None or grid_constructor is not None:
            raise ValueError("mdgrad_amd integrates on the supplied time grid only")
        self.func, self.y0 = func, y0

    def step_func(self, func, t, dt, y):
        raise NotImplementedError

    def integrate(self, t):
        if not bool((t[1:] > t[:-1]).all()):
            raise AssertionError('t must be strictly increasing or decrasing')
        t = t.type_as(self.y0[0]).to(self.y0[0].device)
        frames = [self.y0]
        y = self.y0
        for k in range(t.shape[0] - 1):
            dy = self.step_func(self.func, t[k], t[k + 1] - t[k], y)
            y = tuple(a + b for a, b in zip(y, dy))
            frames.append(y)
        return tuple(torch.stack([f[i] for f in frames]) for i in range(len(self.y0)))


class RK4(FixedGridODESolver):
    """3/8-rule RK4 step (tinydiffeq.py:88-103)."""
    order = 4

    def step_func(self, func, t, dt, y):
        k1 = func(t, y)
        k2 = func(t + dt / 3, tuple(a + dt * b / 3 for a, b in zip(y, k1)))
        k3 = func(t + dt * 2 / 3, tuple(a + dt * (b / -3 + c) for a, b, c in zip(y, k1, k2)))
        k4 = func(t + dt, tuple(a + dt * (b - c + d) for a, b, c, d in zip(y, k1, k2, k3)))
        return tuple((b + 3 * c + 3 * d + e) * (dt / 8) for b, c, d, e in zip(k1, k2, k3, k4))
