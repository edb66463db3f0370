"""Import name of the package that lives in ``image-preprocessing-pipeline_amd/``.

The directory name required by the repo layout contains hyphens, which Python cannot import;
this shim points the package search path at that directory, so
``import ipp_amd.decon`` loads ``image-preprocessing-pipeline_amd/decon.py``.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "image-preprocessing-pipeline_amd")
__path__ = [_PKG_DIR]
PACKAGE_DIR = _PKG_DIR
