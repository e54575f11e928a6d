"""
cost.py - the Cost plugin base class.

Same contract as qoc/models/cost.py:5-51: class attributes `name`,
`requires_step_evaluation`, instance attribute `cost_multiplier`, and
`cost(controls, states, system_eval_step) -> scalar`, called with the WHOLE control array, the
current (state_count x hilbert_size x 1) states and the integer system step.

The reference differentiates `cost` with autograd. This package has no AD engine; the
derivative of a cost reaches the engine in one of these ways:

* `device_descriptor(state_count, hilbert_size)` -> dict for the HIP engine (built-in state
  costs: the value AND the cotangent are evaluated on the GPU);
* `controls_bar(controls, states, system_eval_step)` -> d cost / d Re(controls) + i d cost /
  d Im(controls) for costs that depend on the controls only (`uses_states = False`);
* any other subclass (a user's own cost of the states or densities) works unchanged in both
  GRAPE entry points: the host evaluates `cost()` on the states the device returns and hands the engine the
  cotangent of the states at every cost step - from the optional hook
  `states_bar(controls, states, system_eval_step)` (d cost / d Re(states) + i d cost /
  d Im(states)), else by central differences of `cost()`. The explicit dependence on the
  controls is differentiated the same way (`controls_bar()` hook, or `uses_controls = False`
  to declare there is none). The finite-difference route costs 4 * state_count * hilbert_size
  evaluations of `cost()` per cost step (x hilbert_size for densities) and is accurate to
  ~1e-8; write the hooks for speed.
"""


class Cost(object):
    name = "parent_cost"
    requires_step_evaluation = False
    uses_states = True
    uses_controls = True

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

    def states_bar(self, controls, states, system_eval_step):
        return None
