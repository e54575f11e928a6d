"""
adam.py - the Adam optimizer plugin (arXiv:1412.6980).

Same constructor, attributes, `run(function, iteration_count, initial_params, jacobian, args)`
contract and `update(grads, params)` arithmetic as qoc/standard/optimizers/adam.py:9-165:
`run` never calls `function`; `jacobian(params, *args) -> (grads, terminate)`.
"""

import numpy as np

from qoc_amd.models.policies import OperationPolicy


class Adam(object):
    name = "adam"

    def __init__(self, beta_1=0.9, beta_2=0.999, clip_grads=None, epsilon=1e-8,
                 learning_rate=1e-3, learning_rate_decay=None,
                 operation_policy=OperationPolicy.CPU, scale_grads=None):
        super().__init__()
        self.apply_scale_grads = scale_grads is not None
        self.apply_clip_grads = clip_grads is not None
        self.apply_learning_rate_decay = learning_rate_decay is not None
        self.beta_1 = beta_1
        self.beta_2 = beta_2
        self.clip_grads = clip_grads
        self.epsilon = epsilon
        self.gradient_moment = None
        self.gradient_square_moment = None
        self.initial_learning_rate = learning_rate
        self.iteration_count = 0
        self.learning_rate = learning_rate
        self.learning_rate_decay = learning_rate_decay
        self.scale_grads = scale_grads

    def __str__(self):
        return ("{}, beta_1: {}, beta_2: {}, epsilon: {}, lr0: {}, "
                "lr_decay: {}, clip_grads: {}, scale_grads: {}"
                "".format(self.name, self.beta_1, self.beta_2, self.epsilon,
                          self.initial_learning_rate, self.learning_rate_decay,
                          self.clip_grads, self.scale_grads))

    def run(self, function, iteration_count, initial_params, jacobian, args=()):
        self.iteration_count = 0
        self.gradient_moment = np.zeros_like(initial_params)
        self.gradient_square_moment = np.zeros_like(initial_params)
        params = initial_params
        for _ in range(iteration_count):
            grads, terminate = jacobian(params, *args)
            if terminate:
                break
            params = self.update(grads, params)

    def update(self, grads, params):
        if self.apply_learning_rate_decay:
            learning_rate = (self.initial_learning_rate
                             * np.exp(-np.divide(self.iteration_count, self.learning_rate_decay)))
        else:
            learning_rate = self.initial_learning_rate
        if self.apply_scale_grads:
            grads = (grads / np.linalg.norm(grads)) * self.scale_grads
        if self.apply_clip_grads:
            grads = np.clip(grads, -self.clip_grads, self.clip_grads)
        self.iteration_count += 1
        step = self.iteration_count
        self.gradient_moment = self.beta_1 * self.gradient_moment + (1 - self.beta_1) * grads
        self.gradient_square_moment = (self.beta_2 * self.gradient_square_moment
                                       + (1 - self.beta_2) * np.square(grads))
        moment_hat = np.divide(self.gradient_moment, 1 - np.power(self.beta_1, step))
        square_hat = np.divide(self.gradient_square_moment, 1 - np.power(self.beta_2, step))
        return params - learning_rate * np.divide(moment_hat, np.sqrt(square_hat) + self.epsilon)
