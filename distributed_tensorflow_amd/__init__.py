"""MI355X-native parameter-server training framework.

Same capabilities as the TF-1.x script /root/reference/distributed.py (ps/worker
roles, synchronous SyncReplicas-style and asynchronous Hogwild-style updates,
Supervisor-style init/restore/checkpoint, MNIST MLP training loop), re-designed
for one process per MI355X: HIP/CDNA4 kernels for the hot path, parameter and
optimizer state sharded in HBM, RCCL collectives over xGMI.
"""
__version__ = "0.1.0"
