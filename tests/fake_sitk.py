"""npy-backed stand-in for the four SimpleITK calls the T2-mapping driver makes (tests only; the
image has no SimpleITK).  ReadImage(path) loads path + '.npy'; WriteImage keeps images in `written`."""
import sys
import types

import numpy as np


class Image:
    def __init__(self, arr, spacing=(1.0, 1.0, 1.5), origin=(-3.0, 4.0, 5.0), direction=(1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0)):
        self.arr, self.spacing, self.origin, self.direction = np.asarray(arr), tuple(spacing), tuple(origin), tuple(direction)

    def GetSpacing(self): return self.spacing
    def GetOrigin(self): return self.origin
    def GetDirection(self): return self.direction
    def SetSpacing(self, s): self.spacing = tuple(s)
    def SetOrigin(self, o): self.origin = tuple(o)
    def SetDirection(self, d): self.direction = tuple(d)


def install():
    m = types.ModuleType("SimpleITK")
    m.written = {}
    m.Image = Image
    m.ReadImage = lambda path: Image(np.load(path + ".npy"))
    m.GetArrayFromImage = lambda img: img.arr
    m.GetImageFromArray = lambda arr: Image(arr, (1, 1, 1), (0, 0, 0))
    m.WriteImage = lambda img, path: m.written.__setitem__(path, img)
    sys.modules["SimpleITK"] = m
    return m
