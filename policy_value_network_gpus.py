#coding:utf-8
"""Drop-in for the reference's policy_value_network_gpus.py (`--processor gpu`).

The reference builds `num_gpus` in-graph towers in ONE process, keeps the variables on /cpu:0 and
averages tower gradients through host memory (policy_value_network_gpus.py:66-95,206-250).  The
MI355X-native scheme is one process per GPU (torch.distributed over RCCL/xGMI): every rank holds a
replica and plays its own shard of games; gradients are all-reduced (cchess_zero_amd/parallel.py).
So inside one process this class IS the single-device network on the local GPU; `num_gpus` is
recorded for the launcher (python -m torch.distributed.run --nproc-per-node <num_gpus> main.py ...).
forward() keeps the reference's signature; the remainder-padding of the reference's tower split
(:323-330) is unnecessary because the batch is never split inside a process.
"""
from policy_value_network import policy_value_network


class policy_value_network_gpus(policy_value_network):
    def __init__(self, num_gpus=1, res_block_nums=7, **kw):
        kw.setdefault("save_dir", "./gpu_models")   # policy_value_network_gpus.py:14
        super().__init__(res_block_nums, **kw)
        self.num_gpus = num_gpus
