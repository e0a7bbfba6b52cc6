"""The two host helpers of the reference's src/helpers/utils.py that the hot path touches."""


def get_scheduled_params(param, param_schedule, step_counter, ignore_schedule=False):
    """utils.py:64-72: value scaled by vals[idx], idx = first schedule boundary that exceeds step_counter."""
    if ignore_schedule is False:
        vals, steps = param_schedule['vals'], param_schedule['steps']
        assert len(vals) == len(steps) + 1, f'Mispecified schedule! - {param_schedule}'
        idx = len(steps)
        for i, s in enumerate(steps):
            if step_counter < s:
                idx = i
                break
        param = param * vals[idx]
    return param


class Struct:
    def __init__(self, **entries):
        self.__dict__.update(entries)
