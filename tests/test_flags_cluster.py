"""Flag surface (distributed.py:10-32), role validation (:40-47), cluster spec (:49-53), placement (:60-64)."""
import pytest

from distributed_tensorflow_amd import flags as F
from distributed_tensorflow_amd.cluster.spec import (ClusterSpec, Role, round_robin_placement, split_hosts, validate_role)


def _fv():
    fv = F.FlagValues()
    F.define_reference_flags(fv)
    return fv


def test_reference_flag_names_types_defaults():
    fv = _fv()
    assert fv.defaults_dict() == {
        'data_dir': '/home/zhangzhaoyu/incubator-mxnet-master/example/image-classification/data',
        'hidden_units': 100, 'train_steps': 100000, 'batch_size': 100, 'learning_rate': 0.01,
        'ps_hosts': '172.16.1.182:2222',
        'worker_hosts': '172.16.1.183:2223,172.16.1.183:2224,172.16.1.187:2225,172.16.1.187:2226',
        'job_name': None, 'task_index': None, 'sync_replicas': False, 'replicas_to_aggregate': None}
    h = fv.help_dict()
    assert h['job_name'] == 'job name: worker or ps'
    assert h['batch_size'] == 'Training batch size '
    assert h['replicas_to_aggregate'].endswith('default:num_workers)')


def test_flag_syntaxes():
    fv = _fv()
    rest = fv.parse(["--job_name=worker", "--task_index", "3", "--sync_replicas", "--learning_rate=0.5", "extra",
                     "-hidden_units=7", "--replicas_to_aggregate=2"])
    assert rest == ["extra"]
    assert (fv.job_name, fv.task_index, fv.sync_replicas, fv.learning_rate, fv.hidden_units, fv.replicas_to_aggregate) == \
        ("worker", 3, True, 0.5, 7, 2)
    fv.reset()
    fv.parse(["--nosync_replicas"])
    assert fv.sync_replicas is False
    fv.parse(["--sync_replicas=true"])
    assert fv.sync_replicas is True
    with pytest.raises(F.FlagError):
        fv.parse(["--no_such_flag=1"])
    with pytest.raises(F.FlagError):
        fv.parse(["--task_index=abc"])
    with pytest.raises(F.FlagError):
        fv.parse(["--task_index"])


def test_missing_role_raises_value_error_with_reference_messages():
    with pytest.raises(ValueError, match='Must specify an explicit job_name !'):
        validate_role(None, 0)
    with pytest.raises(ValueError, match='Must specify an explicit job_name !'):
        validate_role('', 0)
    with pytest.raises(ValueError, match='Must specify an explicit task_index!'):
        validate_role('worker', None)
    validate_role('ps', 0)
    # the job name is echoed before the task index is examined (distributed.py:40-47)
    said = []
    with pytest.raises(ValueError):
        validate_role('worker', None, echo=said.append)
    assert said == ['job_name : worker']
    said.clear()
    validate_role('worker', 3, echo=said.append)
    assert said == ['job_name : worker', 'task_index : 3']


def test_cluster_spec_and_roles():
    fv = _fv()
    cluster = ClusterSpec({'ps': split_hosts(fv.ps_hosts), 'worker': split_hosts(fv.worker_hosts)})
    assert cluster.num_tasks('worker') == 4 and cluster.num_tasks('ps') == 1
    assert cluster.task_address('worker', 2) == '172.16.1.187:2225'
    r = Role('worker', 0, cluster)
    assert r.is_chief and r.num_workers == 4 and r.worker_device == "/job:worker/task:0/gpu:0"
    assert not Role('worker', 1, cluster).is_chief
    assert not Role('ps', 0, cluster).is_chief
    # two workers per node, as in the README launch lines: local GPU index 0/1 on each host
    assert [Role('worker', i, cluster).local_rank() for i in range(4)] == [0, 1, 0, 1]
    with pytest.raises(ValueError):
        cluster.task_address('worker', 4)
    with pytest.raises(ValueError):
        ClusterSpec({'worker': ['nohostport']})


def test_round_robin_variable_placement():
    names = ["global_step", "hid_w", "hid_b", "sm_w", "sm_b"]        # creation order, distributed.py:65-73
    assert round_robin_placement(names, 1) == {n: 0 for n in names}
    assert round_robin_placement(names, 2) == {"global_step": 0, "hid_w": 1, "hid_b": 0, "sm_w": 1, "sm_b": 0}
    assert round_robin_placement(names, 3)["sm_w"] == 0


def test_integers_are_decimal_unless_prefixed():
    from distributed_tensorflow_amd import flags as F
    fv = F.FlagValues()
    F.DEFINE_integer("task_index", None, "", fv)
    F.DEFINE_integer("mask", 0, "", fv)
    fv.parse(["--task_index=01", "--mask=0x10"])
    assert fv.task_index == 1 and fv.mask == 16
    import pytest
    with pytest.raises(F.FlagError):
        fv.parse(["--task_index=1.5"])


def test_auto_backend_rule_and_shuffle_seed():
    from distributed_tensorflow_amd import train
    assert train.local_hosts(["127.0.0.1:2223", "localhost:2224"])
    assert not train.local_hosts(["127.0.0.1:2223", "172.16.1.183:2224"])          # a worker elsewhere: ps processes needed
    assert train.shuffle_seed(7, 0) != train.shuffle_seed(7, 1) and train.shuffle_seed(7, 1) == train.shuffle_seed(7, 1)
    assert train.shuffle_seed(0, 0) >= 0                                             # from entropy, never the fixed rank
