"""YAML -> nested config dict with the reference's defaults and merge rules (configs/parser.py:6-133)."""
import torch
import yaml


class YAMLParser:
    def __init__(self, config):
        self.reset_config()
        self.parse_config(config)
        self.get_device()
        self.init_seeds()

    config = property(lambda self: self._config)
    device = property(lambda self: self._device)
    loader_kwargs = property(lambda self: self._loader_kwargs)

    def reset_config(self):
        self._config = {
            "experiment": "Default",
            "data": {"mode": "events", "window": 5000},
            "loader": {"resolution": [180, 240], "batch_size": 1, "augment": [], "gpu": 0, "seed": 0},
            "hot_filter": {"enabled": True, "max_px": 100, "min_obvs": 5, "max_rate": 0.8},
            "model": {}, "spiking_neuron": {}, "vis": {"bars": False},
        }

    def parse_config(self, file):
        with open(file) as fid:
            self.parse_dict(yaml.load(fid, Loader=yaml.FullLoader))

    def update(self, config):
        self.reset_config()
        self.parse_config(config)

    def parse_dict(self, input_dict, parent=None):
        parent = self._config if parent is None else parent
        for key, val in input_dict.items():
            if isinstance(val, dict):
                self.parse_dict(val, parent.setdefault(key, {}))
            else:
                parent[key] = val

    def get_device(self):
        gpu = self._config["loader"]["gpu"]
        cuda = False if gpu == 1000 else torch.cuda.is_available()          # gpu: 1000 forces CPU
        self._device = torch.device(("cuda:" + str(gpu) if type(gpu) == int else "cuda:0") if cuda else "cpu")
        self._loader_kwargs = {"num_workers": 0, "pin_memory": True} if cuda else {}

    def init_seeds(self):
        torch.manual_seed(self._config["loader"]["seed"])

    def merge_configs(self, run):
        """MLflow run params (strings; dicts were stringified) overwritten by the current config."""
        config = {k: (eval(v) if len(v) > 0 and v[0] == "{" else v) for k, v in run.items()}
        self.parse_dict(self._config, config)
        return self.combine_entries(config)

    @staticmethod
    def combine_entries(config):
        if "spiking_neuron" in config:
            config["model"]["spiking_neuron"] = config.pop("spiking_neuron")
        return config
