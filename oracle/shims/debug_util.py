"""Stand-in for the reference's missing `debug_util` module: visualisation no-ops."""


def _noop(*a, **k):
    return None


save_modules = viz_result_batch_base = viz_result_batch_goalpred = viz_result_batch_ood = _noop
viz_result_batch_ood_load = viz_data_goal = _noop


def __getattr__(name):
    return _noop


__all__ = ["save_modules", "viz_result_batch_base", "viz_result_batch_goalpred", "viz_result_batch_ood",
           "viz_result_batch_ood_load", "viz_data_goal"]
