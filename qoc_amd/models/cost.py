"""
cost.py - the Cost plugin base class.

Same contract as qoc/models/cost.py:5-51: class attributes `name`,
`requires_step_evaluation`, instance attribute `cost_multiplier`, and
`cost(controls, states, system_eval_step) -> scalar`, called with the WHOLE control array, the
current (state_count x hilbert_size x 1) states and the integer system step.

The reference differentiates `cost` with autograd. This package has no AD engine, so a cost
that takes part in GRAPE also describes its derivative in one of two ways:

* `device_descriptor(state_count, hilbert_size)` -> dict for the HIP engine (built-in state
  costs: the value AND the cotangent are then evaluated on the GPU), or
* `controls_bar(controls, states, system_eval_step)` -> d cost / d Re(controls) + i d cost /
  d Im(controls) for costs that depend on the controls only (`uses_states = False`).

A user subclass that provides neither can be used with evolve_* (forward only); grape_*
rejects it with a clear error.
"""


class Cost(object):
    name = "parent_cost"
    requires_step_evaluation = False
    uses_states = True

    def __init__(self, cost_multiplier=1.):
        super().__init__()
        self.cost_multiplier = cost_multiplier

    def __str__(self):
        return self.name

    def __repr__(self):
        return self.__str__()

    def cost(self, controls, states, system_eval_step):
        raise NotImplementedError("The cost {} has not implemented an evaluation function."
                                  "".format(self))

    # -- hooks used instead of autograd --------------------------------------------------------
    def device_descriptor(self, state_count, hilbert_size):
        return None

    def controls_bar(self, controls, states, system_eval_step):
        return None
