"""Minimal logger shim: the hot path only needs `log` (load_state_dict_ messages, reference unet:1042-1052).
The reference's OpenAI-baselines logger (logger.py:1-496) is observability, out of the hot-path scope."""
import sys

_quiet = False


def configure(dir=None, format_strs=None, comm=None, log_suffix=""):
    return None


def set_quiet(q=True):
    global _quiet
    _quiet = q


def log(*args):
    if not _quiet:
        print(*args, file=sys.stderr)


info = warn = error = debug = log


def logkv(key, val):
    pass


def logkv_mean(key, val):
    pass


def dumpkvs():
    return {}


def get_dir():
    return None
