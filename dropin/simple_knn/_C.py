"""Mirror of ``simple_knn._C`` (submodules/simple-knn/ext.cpp:15-17)."""
from instascene_amd.knn import distCUDA2  # noqa: F401
