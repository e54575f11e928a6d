"""
common.py - control-array plumbing shared by the entry points.

Behaviour follows qoc/core/common.py: in-place modulus clipping (:8-30), the three initial
control generators (:33-142), initial-control validation and defaults (:146-198) and the
optimizer <-> cost-function layouts (:201-246): a complex (Nc x K) array travels through the
optimizer as [Re(ravel) ..., Im(ravel) ...].
"""

import numpy as np


def clip_control_norms(controls, max_control_norms):
    """Rescale, IN PLACE, every entry of column i whose modulus exceeds max_control_norms[i]."""
    for i, max_norm in enumerate(max_control_norms):
        column = controls[:, i]
        moduli = np.abs(column)
        over = np.nonzero(np.less(max_norm, moduli))
        column[over] = (column[over] / moduli[over]) * max_norm


def _complexify(controls, complex_controls):
    if complex_controls:
        return (controls - 1j * controls) / np.sqrt(2)
    return controls


def gen_controls_cos(complex_controls, control_count, control_eval_count, evolution_time,
                     max_control_norms, periods=10.):
    """Cosine of amplitude max/2; exact zeros are replaced by max/10."""
    omega = np.divide(2 * np.pi, np.divide(control_eval_count, periods))
    controls = np.zeros((control_eval_count, control_count))
    for i in range(control_count):
        max_norm = max_control_norms[i]
        wave = np.divide(max_norm, 2) * np.cos(omega * np.arange(control_eval_count))
        controls[:, i] = np.where(wave, wave, max_norm * 1e-1)
    return _complexify(controls, complex_controls)


def gen_controls_white(complex_controls, control_count, control_eval_count, evolution_time,
                       max_control_norms, periods=10.):
    """White noise of standard deviation max/5 (unseeded, as in the reference)."""
    controls = np.zeros((control_eval_count, control_count))
    for i in range(control_count):
        controls[:, i] = np.random.normal(0, max_control_norms[i] / 5.0, control_eval_count)
    return _complexify(controls, complex_controls)


def gen_controls_flat(complex_controls, control_count, control_eval_count, evolution_time,
                      max_control_norms, periods=10.):
    """Flat line at max/10."""
    controls = np.zeros((control_eval_count, control_count))
    for i in range(control_count):
        controls[:, i] = np.repeat(max_control_norms[i] * 1e-1, control_eval_count)
    return _complexify(controls, complex_controls)


_NORM_TOLERANCE = 1e-10


def initialize_controls(complex_controls, control_count, control_eval_count, evolution_time,
                        initial_controls, max_control_norms):
    """Defaults (norms = 1, flat controls) and validation of user supplied initial controls."""
    if max_control_norms is None:
        max_control_norms = np.ones(control_count)
    if initial_controls is None:
        controls = gen_controls_flat(complex_controls, control_count, control_eval_count,
                                     evolution_time, max_control_norms)
        return controls, max_control_norms
    if complex_controls and not np.iscomplexobj(initial_controls):
        raise ValueError("The program expected that the initial_controls specified by "
                         "the user conformed to complex_controls, but "
                         "the program found that the initial_controls were not complex "
                         "and complex_controls was set to True.")
    if not complex_controls and np.iscomplexobj(initial_controls):
        raise ValueError("The program expected that the initial_controls specified by "
                         "the user conformed to complex_controls, but "
                         "the program found that the initial_controls were complex "
                         "and complex_controls was set to False.")
    for control_step, step_controls in enumerate(initial_controls):
        if not np.less_equal(np.abs(step_controls), max_control_norms + _NORM_TOLERANCE).all():
            raise ValueError("The program expected that the initial_controls specified by "
                             "the user conformed to max_control_norms, but the program "
                             "found a conflict at initial_controls[{}]={} and "
                             "max_control_norms={}."
                             "".format(control_step, step_controls, max_control_norms))
    return initial_controls, max_control_norms


def slap_controls(complex_controls, controls, controls_shape):
    """Optimizer format -> cost-function format. Real controls stay a VIEW of the input."""
    if complex_controls:
        real, imag = np.split(controls, 2)
        controls = real + 1j * imag
    return np.reshape(controls, controls_shape)


def strip_controls(complex_controls, controls):
    """Cost-function format -> optimizer format."""
    controls = np.ravel(controls)
    if complex_controls:
        controls = np.hstack((np.real(controls), np.imag(controls)))
    return controls
