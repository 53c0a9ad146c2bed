"""Training callbacks: early stopping, evaluation logging / recording, parameter schedules.

``early_stopping_rounds = 5`` together with ``verbose = 0L`` is how every CV call
of the reference runs (r/gridsearchCV.R:77-78, 114-115).

Compatibility note: the callback PROTOCOL here is the one of the public LightGBM Python package - the fields of
``CallbackEnv``, ``EarlyStopException``, the names and arguments of ``early_stopping`` / ``log_evaluation`` /
``record_evaluation`` / ``reset_parameter``, the ``order`` / ``before_iteration`` attributes and the wording of the log
lines - so that callbacks and log parsers written against that package work unchanged with this one.  The structure
of the functions follows from that protocol; nothing in /root/reference contains such code.
"""
from __future__ import annotations

import collections
import logging
from typing import Any, Callable, Dict, List, Union

from .basic import _log_info

logger = logging.getLogger("lightgbm_amd")

CallbackEnv = collections.namedtuple(
    "CallbackEnv", ["model", "params", "iteration", "begin_iteration", "end_iteration", "evaluation_result_list"])


class EarlyStopException(Exception):
    """Raised by a callback to stop training."""

    def __init__(self, best_iteration: int, best_score):
        super().__init__()
        self.best_iteration = best_iteration
        self.best_score = best_score


def _format_eval_result(value, show_stdv: bool = True) -> str:
    if len(value) == 4:
        return f"{value[0]}'s {value[1]}: {value[2]:g}"
    if len(value) == 5:
        if show_stdv:
            return f"{value[0]}'s {value[1]}: {value[2]:g} + {value[4]:g}"
        return f"{value[0]}'s {value[1]}: {value[2]:g}"
    raise ValueError("Wrong metric value")


def log_evaluation(period: int = 1, show_stdv: bool = True) -> Callable:
    """Logs the evaluation results every ``period`` iterations."""

    def _callback(env: CallbackEnv) -> None:
        if period > 0 and env.evaluation_result_list and (env.iteration + 1) % period == 0:
            result = "\t".join(_format_eval_result(x, show_stdv) for x in env.evaluation_result_list)
            logger.info("[%d]\t%s", env.iteration + 1, result)
            _log_info(f"[{env.iteration + 1}]\t{result}")

    _callback.order = 10
    return _callback


print_evaluation = log_evaluation


def record_evaluation(eval_result: Dict[str, Dict[str, List[Any]]]) -> Callable:
    """Records the evaluation history into ``eval_result`` (``{data: {metric: [values]}}``)."""
    if not isinstance(eval_result, dict):
        raise TypeError("eval_result should be a dictionary")

    def _init(env: CallbackEnv) -> None:
        eval_result.clear()
        for item in env.evaluation_result_list:
            eval_result.setdefault(item[0], collections.OrderedDict())
            if len(item) == 4:
                eval_result[item[0]].setdefault(item[1], [])
            else:
                eval_result[item[0]].setdefault(f"{item[1]}-mean", [])
                eval_result[item[0]].setdefault(f"{item[1]}-stdv", [])

    def _callback(env: CallbackEnv) -> None:
        if env.iteration == env.begin_iteration:
            _init(env)
        for item in env.evaluation_result_list:
            if len(item) == 4:
                eval_result[item[0]][item[1]].append(item[2])
            else:
                eval_result[item[0]][f"{item[1]}-mean"].append(item[2])
                eval_result[item[0]][f"{item[1]}-stdv"].append(item[4])

    _callback.order = 20
    return _callback


def reset_parameter(**kwargs: Union[list, Callable]) -> Callable:
    """Resets parameters after the first iteration; values are lists (per iteration) or functions of the iteration."""

    def _callback(env: CallbackEnv) -> None:
        new_parameters = {}
        for key, value in kwargs.items():
            if isinstance(value, list):
                if len(value) != env.end_iteration - env.begin_iteration:
                    raise ValueError(f"Length of list {key!r} has to equal 'num_boost_round'.")
                new_param = value[env.iteration - env.begin_iteration]
            else:
                new_param = value(env.iteration - env.begin_iteration)
            if new_param != env.params.get(key, None):
                new_parameters[key] = new_param
        if new_parameters:
            env.model.reset_parameter(new_parameters)
            env.params.update(new_parameters)

    _callback.before_iteration = True
    _callback.order = 10
    return _callback


def early_stopping(stopping_rounds: int, first_metric_only: bool = False, verbose: bool = True,
                   min_delta: float = 0.0) -> Callable:
    """Stops when no validation metric improved in the last ``stopping_rounds`` rounds.

    The training data's own metrics are ignored.  At the last scheduled iteration the
    best iteration found so far is reported as well, so ``best_iteration`` is always set.
    """
    best_score: List[float] = []
    best_iter: List[int] = []
    best_score_list: List[Any] = []
    cmp_op: List[Callable] = []
    enabled = [True]
    first_metric = [""]

    def _init(env: CallbackEnv) -> None:
        enabled[0] = bool(env.evaluation_result_list)
        if not enabled[0]:
            logger.warning("Early stopping is not available without validation data")
            return
        if verbose:
            _log_info(f"Training until validation scores don't improve for {stopping_rounds} rounds")
        first_metric[0] = env.evaluation_result_list[0][1].split(" ")[-1]
        best_score.clear(); best_iter.clear(); best_score_list.clear(); cmp_op.clear()
        for item in env.evaluation_result_list:
            best_iter.append(0)
            best_score_list.append(None)
            if item[3]:   # higher is better
                best_score.append(float("-inf"))
                cmp_op.append(lambda a, b: a > b + min_delta)
            else:
                best_score.append(float("inf"))
                cmp_op.append(lambda a, b: a < b - min_delta)

    def _final_iteration_check(env: CallbackEnv, i: int) -> None:
        if env.iteration == env.end_iteration - 1:
            if verbose:
                _log_info("Did not meet early stopping. Best iteration is:\n[%d]\t%s" % (
                    best_iter[i] + 1, "\t".join(_format_eval_result(x) for x in best_score_list[i])))
            raise EarlyStopException(best_iter[i], best_score_list[i])

    def _callback(env: CallbackEnv) -> None:
        if env.iteration == env.begin_iteration:
            _init(env)
        if not enabled[0]:
            return
        for i, item in enumerate(env.evaluation_result_list):
            score = item[2]
            if best_score_list[i] is None or cmp_op[i](score, best_score[i]):
                best_score[i] = score
                best_iter[i] = env.iteration
                best_score_list[i] = env.evaluation_result_list
            metric_name = item[1].split(" ")[-1]
            if first_metric_only and first_metric[0] != metric_name:
                continue
            train_name = getattr(env.model, "_train_data_name", "training")
            if item[0] == train_name or (item[0] == "cv_agg" and item[1].split(" ")[0] == "train"):
                _final_iteration_check(env, i)
                continue
            if env.iteration - best_iter[i] >= stopping_rounds:
                if verbose:
                    _log_info("Early stopping, best iteration is:\n[%d]\t%s" % (
                        best_iter[i] + 1, "\t".join(_format_eval_result(x) for x in best_score_list[i])))
                raise EarlyStopException(best_iter[i], best_score_list[i])
            _final_iteration_check(env, i)

    _callback.order = 30
    return _callback
