"""
lbfgsb.py - SciPy's L-BFGS-B as an optimizer plugin (qoc/standard/optimizers/lbfgsb.py:7-49):
calls both `function` and `jacobian`, ignores their terminate flags, maxiter = iteration_count.
"""

from scipy.optimize import minimize


class LBFGSB(object):
    def __init__(self):
        super().__init__()

    def run(self, function, iteration_count, initial_params, jacobian, args=()):
        def value(*a, **kw):
            return function(*a, **kw)[0]

        def gradient(*a, **kw):
            return jacobian(*a, **kw)[0]

        return minimize(value, initial_params, args=args, method="L-BFGS-B", jac=gradient,
                        options={"maxiter": iteration_count})
