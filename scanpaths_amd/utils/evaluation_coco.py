"""Import surface of COCO_Search18/utils/evaluation.py (human_evaluation :11, evaluation :180, pairs_eval_scanmatch :313) on the batched
device scorers of scanpaths_amd.utils.evaluation:   from scanpaths_amd.utils.evaluation_coco import human_evaluation, evaluation, pairs_eval_scanmatch"""
from functools import partial

from .evaluation import evaluation, human_evaluation_free_viewing, pairs_eval_scanmatch      # noqa: F401

human_evaluation = partial(human_evaluation_free_viewing, task="COCO_Search18")
