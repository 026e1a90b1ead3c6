"""``lib/networks/enerf/network_human_amd.py``: the ZJU-MoCap / ``gui_human.py`` variant (replaces
``lib/networks/enerf/network_human.py``; ``network_module lib.networks.enerf.network_human_amd``)."""
from lib.config import cfg
from enerf_amd.config import EnerfConfig
from enerf_amd.network import NetworkHuman as _AmdNetwork


class Network(_AmdNetwork):
    def __init__(self):
        super().__init__(EnerfConfig.from_yacs(cfg))
