"""Parallel training engines: sharded parameter-server state, synchronous
(SyncReplicas) and asynchronous (Hogwild push/pull) update modes."""
