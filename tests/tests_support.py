"""Small fakes shared by the GPU tests (a GaussianModel-like object with the reference getters)."""


class _PC:
    def __init__(self, inp, active_sh_degree=3):
        self._i = {k: (v.cuda() if v is not None else None) for k, v in inp.items()}
        self.active_sh_degree = active_sh_degree

    get_xyz = property(lambda s: s._i["means3D"])
    get_opacity = property(lambda s: s._i["opacities"])
    get_scaling = property(lambda s: s._i["scales"])
    get_rotation = property(lambda s: s._i["rotations"])
    get_features = property(lambda s: s._i["shs"])
    get_seg_feature = property(lambda s: s._i["extra"])


def pc_from_inputs(inp):
    return _PC(inp)
