"""autograd.wrap_util stand-in."""


def unary_to_nary(operator):
    return operator
