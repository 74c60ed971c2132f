"""
gcn-over-pruned-trees_amd: MI355X-native (gfx950) implementation of ONE hot path of
gstoica27/gcn-over-pruned-trees -- the pruned-dependency-tree adjacency build
(reference model/tree.py:58-204) and the degree-normalised masked-adjacency GCN layer loop
(reference model/gcn.py:258-395, `adj_type == 'regular'`), forward and backward.

Layout
  csrc/    hand-written HIP kernels + the C-ABI (include/gcnpt.h) -> csrc/libgcnpt.so
  _lib.py  ctypes binding of that C-ABI (fails loudly when the library is missing)
  model/   host-side mirror of the reference's interface for this path
           (model.tree, model.gcn: GCN / GCNRelationModel / GCNClassifier / pool)
  shard.py sentence sharding + flat-bucket gradient all-reduce for data parallel runs
  utils/   restated constants and the synthetic TACRED-shaped batch generator

Import as `gcn_over_pruned_trees_amd` (alias module at the repo root).
"""
__version__ = "0.1.0"
