"""Drop-in for the reference's ``gaussian_renderer`` package (``render`` only; the SIBR network GUI
of gaussian_renderer/network_gui.py is out of scope)."""
from instascene_amd.render import render  # noqa: F401
