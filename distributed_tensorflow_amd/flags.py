"""Command-line flag system with the surface of ``tf.app.flags``.

The eleven flags of /root/reference/distributed.py:10-32 keep their names,
types, defaults and help strings; ``run(main)`` plays the role of
``tf.app.run()`` (distributed.py:167-168): parse ``sys.argv``, then call
``main(unparsed_argv)``.  Framework-specific flags (device, optimizer, model,
staleness bound, checkpoint directory, ...) are defined next to them.

Accepted syntaxes (absl conventions): ``--name=value``, ``--name value``,
``--bool_flag`` / ``--nobool_flag`` / ``--bool_flag=false``; single-dash
spellings are accepted too.
"""
from __future__ import annotations

import sys
from typing import Any, Callable, Dict, List, Optional


class FlagError(ValueError):
    pass


class _Flag:
    __slots__ = ("name", "default", "help", "kind", "value", "present")

    def __init__(self, name, default, help_, kind):
        self.name, self.default, self.help, self.kind = name, default, help_, kind
        self.value = default
        self.present = False

    def parse(self, text: str) -> Any:
        if self.kind == "string":
            return text
        if self.kind == "integer":
            try:
                # decimal unless the value carries a 0x / 0o / 0b prefix: "--task_index=01" is 1, as absl reads it
                t = text.strip()
                body = t.lstrip("+-").lower()
                return int(t, 0) if body[:2] in ("0x", "0o", "0b") else int(t, 10)
            except ValueError:
                raise FlagError(f"flag --{self.name}={text!r}: invalid integer")
        if self.kind == "float":
            try:
                return float(text)
            except ValueError:
                raise FlagError(f"flag --{self.name}={text!r}: invalid float")
        if self.kind == "boolean":
            t = text.strip().lower()
            if t in ("1", "t", "true", "y", "yes"):
                return True
            if t in ("0", "f", "false", "n", "no"):
                return False
            raise FlagError(f"flag --{self.name}={text!r}: invalid boolean")
        raise AssertionError(self.kind)


class FlagValues:
    """Registry + parsed values; attribute access returns the flag's value."""

    def __init__(self):
        object.__setattr__(self, "_flags", {})
        object.__setattr__(self, "_parsed", False)

    # -- definition
    def _define(self, name, default, help_, kind):
        if name in self._flags:
            raise FlagError(f"flag --{name} defined twice")
        self._flags[name] = _Flag(name, default, help_, kind)

    # -- access
    def __getattr__(self, name):
        flags = object.__getattribute__(self, "_flags")
        if name in flags:
            return flags[name].value
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in self._flags:
            self._flags[name].value = value
        else:
            raise AttributeError(f"unknown flag {name}")

    def __contains__(self, name):
        return name in self._flags

    def is_present(self, name) -> bool:
        return self._flags[name].present

    def flag_values_dict(self) -> Dict[str, Any]:
        return {k: f.value for k, f in self._flags.items()}

    def defaults_dict(self) -> Dict[str, Any]:
        return {k: f.default for k, f in self._flags.items()}

    def help_dict(self) -> Dict[str, str]:
        return {k: f.help for k, f in self._flags.items()}

    def reset(self):
        for f in self._flags.values():
            f.value, f.present = f.default, False
        object.__setattr__(self, "_parsed", False)

    # -- parsing
    def parse(self, argv: List[str]) -> List[str]:
        """Parse ``argv`` (without the program name); returns the non-flag arguments."""
        rest: List[str] = []
        i = 0
        while i < len(argv):
            arg = argv[i]
            i += 1
            if arg == "--":
                rest.extend(argv[i:])
                break
            if not arg.startswith("-") or arg == "-":
                rest.append(arg)
                continue
            body = arg.lstrip("-")
            if body in ("help", "helpfull", "h"):
                print(self.usage())
                raise SystemExit(0)
            name, eq, val = body.partition("=")
            flag = self._flags.get(name)
            if flag is None and not eq and name.startswith("no") and name[2:] in self._flags \
                    and self._flags[name[2:]].kind == "boolean":
                flag = self._flags[name[2:]]
                flag.value, flag.present = False, True
                continue
            if flag is None:
                raise FlagError(f"Unknown command line flag '{name}'")
            if flag.kind == "boolean" and not eq:
                flag.value, flag.present = True, True
                continue
            if not eq:
                if i >= len(argv):
                    raise FlagError(f"Missing value for flag --{name}")
                val = argv[i]
                i += 1
            flag.value, flag.present = flag.parse(val), True
        object.__setattr__(self, "_parsed", True)
        return rest

    def usage(self) -> str:
        lines = ["flags:"]
        for f in self._flags.values():
            lines.append(f"  --{f.name}: {f.help}\n    (default: {f.default!r}; {kind_name(f.kind)})")
        return "\n".join(lines)


def kind_name(kind: str) -> str:
    return {"string": "a string", "integer": "an integer", "float": "a number", "boolean": "a boolean"}[kind]


FLAGS = FlagValues()


def DEFINE_string(name, default, help, flag_values: FlagValues = FLAGS):  # noqa: A002,N802
    flag_values._define(name, default, help, "string")


def DEFINE_integer(name, default, help, flag_values: FlagValues = FLAGS):  # noqa: A002,N802
    flag_values._define(name, default, help, "integer")


def DEFINE_float(name, default, help, flag_values: FlagValues = FLAGS):  # noqa: A002,N802
    flag_values._define(name, default, help, "float")


def DEFINE_boolean(name, default, help, flag_values: FlagValues = FLAGS):  # noqa: A002,N802
    flag_values._define(name, default, help, "boolean")


DEFINE_bool = DEFINE_boolean


def define_reference_flags(fv: FlagValues = FLAGS) -> None:
    """The eleven flags of distributed.py:10-32 (names, defaults and help text unchanged)."""
    DEFINE_string('data_dir', '/home/zhangzhaoyu/incubator-mxnet-master/example/image-classification/data',
                  'Directory  for storing mnist data', fv)
    DEFINE_integer('hidden_units', 100, 'Number of units in the hidden layer of the NN', fv)
    DEFINE_integer('train_steps', 100000, 'Number of training steps to perform', fv)
    DEFINE_integer('batch_size', 100, 'Training batch size ', fv)
    DEFINE_float('learning_rate', 0.01, 'Learning rate', fv)
    DEFINE_string('ps_hosts', '172.16.1.182:2222', 'Comma-separated list of hostname:port pairs', fv)
    DEFINE_string('worker_hosts', '172.16.1.183:2223,172.16.1.183:2224,172.16.1.187:2225,172.16.1.187:2226',
                  'Comma-separated list of hostname:port pairs', fv)
    DEFINE_string('job_name', None, 'job name: worker or ps', fv)
    DEFINE_integer('task_index', None, 'Index of task within the job', fv)
    DEFINE_boolean("sync_replicas", False,
                   "Use the sync_replicas (synchronized replicas) mode, "
                   "wherein the parameter updates from workers are "
                   "aggregated before applied to avoid stale gradients", fv)
    DEFINE_integer("replicas_to_aggregate", None,
                   "Number of replicas to aggregate before parameter "
                   "update is applied (For sync_replicas mode only; "
                   "default:num_workers)", fv)


def define_framework_flags(fv: FlagValues = FLAGS) -> None:
    """Flags that have no counterpart in the reference script."""
    DEFINE_string('device', 'auto', "Compute device: 'auto' (GPU if visible, else CPU: the reference's "
                  "allow_soft_placement), 'gpu' or 'cpu'", fv)
    DEFINE_string('ps_backend', 'auto', "Where parameter-server state lives: 'hbm' (sharded across the worker GPUs, "
                  "collectives over xGMI), 'host' (ps processes run the native TCP parameter service, the reference's "
                  "topology) or 'auto' (hbm when every entry of --worker_hosts names this machine, else host)", fv)
    DEFINE_string('model', 'mlp', "Model: 'mlp' (the reference 784-H-10 network), 'wide_mlp' or 'resnet50'", fv)
    DEFINE_string('optimizer', 'sgd', "Optimizer: 'sgd' (reference) or 'adam'", fv)
    DEFINE_string('loss', 'reference', "'reference' reproduces the script's softmax-applied-twice loss; 'standard' is "
                  "cross-entropy on the logits", fv)
    DEFINE_boolean('synthetic_data', False, 'Use a synthetic MNIST-shaped dataset instead of reading --data_dir', fv)
    DEFINE_integer('synthetic_train_size', 55000, 'Rows of the synthetic training set', fv)
    DEFINE_integer('staleness', 0, 'Async mode: a worker pushes its gradient on every step and refreshes its replica from '
                   'the ps-resident parameters on every (staleness+1)-th; 0 = every step, as the reference (send/recv '
                   'engine: pulls that may still be in flight when the next step starts)', fv)
    DEFINE_integer('staleness_bound', None, 'Async mode on the GPUs of one node: a replica that misses more than this many '
                   'updates of OTHER workers is refreshed with the next push, and a worker more than this many steps ahead '
                   'of the slowest waits for it (default: staleness + 2 x (workers - 1); negative: no bound, plain '
                   'Hogwild as the reference runs it)', fv)
    DEFINE_integer('log_every', 1, 'Print the per-step line every N local steps (reference: every step)', fv)
    DEFINE_integer('steps_per_launch', 0, 'Steps replayed per hipGraph launch (0 = choose automatically)', fv)
    DEFINE_integer('validation_every', 10000, 'Validate every N local steps (reference: 10000, including step 0)', fv)
    DEFINE_string('train_dir', None, 'Checkpoint/log directory (default: a fresh temporary directory, as the reference)', fv)
    DEFINE_integer('save_model_secs', 600, 'Chief checkpoint period in seconds (Supervisor default); 0 disables', fv)
    DEFINE_integer('seed', 0, 'Seed of parameter initialisation and data shuffling (0 = from entropy, as the reference)', fv)
    DEFINE_integer('hidden_layers', 1, 'Number of hidden layers (wide_mlp)', fv)
    DEFINE_boolean('post_update_accuracy', True, "Report the training accuracy of each batch with the parameters AFTER "
                   "the update, as the reference's second session run does; false reports it from the training pass", fv)
    DEFINE_string('dist_backend', 'auto', "torch.distributed backend: 'auto' (nccl = RCCL on GPU, gloo on CPU)", fv)
    DEFINE_string('dtype', 'fp32', "Compute dtype of the wide models: 'fp32' or 'bf16' (fp32 accumulate/master weights)", fv)
    DEFINE_boolean('single_process', False, 'Ignore the cluster flags and train in this process only', fv)


def run(main: Optional[Callable[[List[str]], Any]] = None, argv: Optional[List[str]] = None,
        flag_values: FlagValues = FLAGS):
    """``tf.app.run`` equivalent (distributed.py:168)."""
    argv = list(sys.argv if argv is None else argv)
    try:
        rest = flag_values.parse(argv[1:])
    except FlagError as e:
        sys.stderr.write(f"FATAL Flags parsing error: {e}\n")
        raise SystemExit(1)
    if main is None:
        main = sys.modules["__main__"].main
    sys.exit(main([argv[0]] + rest))
