from setuptools import setup, Extension
from Cython.Build import cythonize
import numpy as np
exts = [Extension(n, [n + ".pyx"], include_dirs=[np.get_include()], extra_compile_args=["-O3", "-ffast-math"]) for n in ("topo_param", "transform", "direction")]
setup(name="refmods", ext_modules=cythonize(exts, language_level=3))
