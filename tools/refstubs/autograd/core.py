"""autograd.core stand-in."""


def make_vjp(*args, **kwargs):
    raise NotImplementedError("forward-only autograd stand-in: no reverse mode")
