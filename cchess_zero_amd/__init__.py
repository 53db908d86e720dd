"""cchess_zero_amd — MI355X-native hot path of cchess-zero (MCTS rollout + move legality + batched net eval).

Layout
  csrc/          hand-written HIP kernels + the C-ABI (include/cchess_hip.h) -> libcchess_hip.so
  _lib.py        ctypes binding of the C-ABI (fails loudly if the library is missing)
  rules.py       batched rules ops (K1-K3) over device tensors
  engine.py      lock-step search engine over G trees
  net.py         residual policy/value network (PyTorch-ROCm; MFMA convs)
The reference-named façade (GameBoard, MCTS_tree, cchess_main, policy_value_network) lives in
main.py / policy_value_network.py at the repository root.
"""
__all__ = ["_lib", "rules", "engine", "net"]
