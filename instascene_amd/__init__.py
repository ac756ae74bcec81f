"""instascene_amd — MI355X-native 2DGS surfel rasterizer + contrastive-feature engine.

The compute path is the C-ABI HIP library ``instascene_amd/csrc`` builds
(``libinstascene_hip.so``, declared in ``include/``); the Python modules here are
the host-side mirror of the reference's operator interface.  Nothing in this
package imports ``oracle/``.
"""
__version__ = "0.1.0"
