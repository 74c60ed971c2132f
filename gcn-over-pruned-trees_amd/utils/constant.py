"""
Constants the hot path needs, restated as values (reference utils/constant.py; the label vocabularies
themselves are data of the reference's loaders and are not reproduced here).
"""
PAD_ID = 0                   # utils/constant.py:8
UNK_ID = 1                   # utils/constant.py:10
DEPREL_FORWARD_BOUND = 42    # utils/constant.py:14  child->parent entry = deprel id + 42
DEPREL_REVERSE_BOUND = 84    # utils/constant.py:16
SELF_LOOP_INDEX = 84         # utils/constant.py:17  value written on the diagonal
INFINITY_NUMBER = 1e12       # utils/constant.py:35  pool() fill value

# embedding table sizes = len() of the reference's id maps (utils/constant.py:25,27,29,33)
N_POS = 47
N_NER = 15
N_DEPREL = 85
N_CLASS_TACRED = 42
