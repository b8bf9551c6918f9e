"""Import surface of OSIE/utils/evaluation.py (human_evaluation :11, evaluation :151, pairs_eval :284) on the batched device scorers of
scanpaths_amd.utils.evaluation:   from scanpaths_amd.utils.evaluation_osie import human_evaluation, evaluation, pairs_eval"""
from functools import partial

from .evaluation import evaluation, human_evaluation_free_viewing, pairs_eval      # noqa: F401

human_evaluation = partial(human_evaluation_free_viewing, task="OSIE")
