"""sgd.py - plain gradient descent plugin (qoc/standard/optimizers/sgd.py:7-59)."""


class SGD(object):
    name = "sgd"

    def __init__(self, learning_rate=1e-3):
        super().__init__()
        self.learning_rate = learning_rate

    def run(self, function, iteration_count, initial_params, jacobian, args=()):
        params = initial_params
        for _ in range(iteration_count):
            grads, terminate = jacobian(params, *args)
            if terminate:
                break
            params = self.update(grads, params)

    def update(self, grads, params):
        return params - self.learning_rate * grads
