"""TEST-ONLY stand-in for `omegaconf` (not installed here): scripts/mapping.py:421 calls OmegaConf.load(path) and then only
reads the result by attribute / key (config.model, config.data.params.common_params.n_fft, config.version).  A YAML load into
attribute-access dicts covers that.  Used by tests/test_mirror.py::test_reference_cli_runs_unchanged_on_the_drop_in."""
import yaml


class _Node(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def _wrap(v):
    if isinstance(v, dict):
        return _Node({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


class OmegaConf:
    @staticmethod
    def load(path):
        with open(path) as f:
            return _wrap(yaml.safe_load(f))
