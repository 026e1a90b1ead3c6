"""Drop this file into the reference tree as ``lib/networks/enerf/network_amd.py`` and select it with
``network_module lib.networks.enerf.network_amd``: ``lib/networks/make_network.py:5-9`` then loads the MI355X-native
renderer behind the reference's own ``Network`` surface (``run.py`` / ``gui_human.py`` / ``net_utils.load_network``
stay unmodified).  Needs this repository on PYTHONPATH and the built ``enerf_amd/libenerf_hip.so``."""
from lib.config import cfg
from enerf_amd.config import EnerfConfig
from enerf_amd.network import Network as _AmdNetwork


class Network(_AmdNetwork):
    def __init__(self):
        super().__init__(EnerfConfig.from_yacs(cfg))
