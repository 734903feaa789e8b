"""Small logger with the call surface the scripts and the train loop use (reference logger.py:1-496 is the OpenAI
baselines logger: observability, out of the hot-path scope): `configure(dir)`, `get_dir()`, `log`/`info`/`warn`,
`logkv`, `logkv_mean`, `dumpkvs` (prints the table and appends it to <dir>/progress.csv), `get_current().name2val`."""
import os
import sys
import tempfile
import time
from collections import defaultdict

_quiet = False


class _Logger:
    def __init__(self, dir=None):
        self.dir = dir
        self.name2val = defaultdict(float)
        self.name2cnt = defaultdict(int)
        self._csv_keys = None

    def logkv(self, k, v):
        self.name2val[k] = v

    def logkv_mean(self, k, v):
        old, cnt = self.name2val[k], self.name2cnt[k]
        self.name2val[k] = old * cnt / (cnt + 1) + float(v) / (cnt + 1)
        self.name2cnt[k] = cnt + 1

    def dumpkvs(self):
        d = dict(self.name2val)
        if d and not _quiet:
            w = max(len(k) for k in d)
            print("\n".join(f"| {k:<{w}} | {d[k]:<12.6g} |" if isinstance(d[k], float) else f"| {k:<{w}} | {d[k]!s:<12} |" for k in sorted(d)),
                  file=sys.stderr)
        if d and self.dir:
            keys = sorted(d)
            path = os.path.join(self.dir, "progress.csv")
            new = self._csv_keys != keys
            with open(path, "a") as f:
                if new:
                    f.write(",".join(keys) + "\n")
                    self._csv_keys = keys
                f.write(",".join(str(d[k]) for k in keys) + "\n")
        self.name2val.clear()
        self.name2cnt.clear()
        return d


_current = _Logger()


def configure(dir=None, format_strs=None, comm=None, log_suffix=""):
    global _current
    if dir is None:
        dir = os.getenv("OPENAI_LOGDIR") or os.path.join(tempfile.gettempdir(), time.strftime("mmd-%Y-%m-%d-%H-%M-%S"))
    os.makedirs(os.path.expanduser(dir), exist_ok=True)
    _current = _Logger(os.path.expanduser(dir))
    return _current


def get_current():
    return _current


def set_quiet(q=True):
    global _quiet
    _quiet = q


def log(*args):
    if not _quiet:
        print(*args, file=sys.stderr)


info = warn = error = debug = log


def logkv(key, val):
    _current.logkv(key, val)


def logkv_mean(key, val):
    _current.logkv_mean(key, val)


def dumpkvs():
    return _current.dumpkvs()


def get_dir():
    return _current.dir
