"""Generates tests/golden/psf_golden.npz by IMPORTING the reference's Python PSF generator
(/root/reference/LsDeconvolveMultiGPU/psf_generator.py, generate_psf :50-121) in the build container.

The reference module imports ``tifffile`` (absent in this image, only used by its ``__main__``): an
empty stand-in module object is registered so the import succeeds (ordinary ModuleNotFoundError, not a
permission denial).  Nothing of the reference's text is stored: only the
arguments and the returned arrays.
    python tests/golden/make_psf_golden.py
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

REF = "/root/reference/LsDeconvolveMultiGPU"

CASES = {
    # name: kwargs of generate_psf
    "em525_ex488": dict(lambda_em=525.0, lambda_ex=488.0, numerical_aperture=0.4, dxy=422.0, dz=1000.0,
                        refractive_index=1.42, f_cylinder_lens=240.0, slit_width=12.0),
    "em642_ex680_default": dict(),
    "em600_ex561_fine": dict(lambda_em=600.0, lambda_ex=561.0, numerical_aperture=0.4, dxy=200.0, dz=600.0,
                             refractive_index=1.42, f_cylinder_lens=240.0, slit_width=12.0),
    "em525_ex488_blur": dict(lambda_em=525.0, lambda_ex=488.0, numerical_aperture=0.4, dxy=422.0, dz=1000.0,
                             refractive_index=1.42, f_cylinder_lens=240.0, slit_width=12.0, gaussian_sgima=0.5),
}


def main():
    stub = types.ModuleType("tifffile")
    stub.imwrite = lambda *a, **k: None
    sys.modules.setdefault("tifffile", stub)
    sys.path.insert(0, REF)
    import psf_generator as ref  # noqa: E402

    out = {"names": np.array(list(CASES))}
    for name, kw in CASES.items():
        with contextlib.redirect_stdout(io.StringIO()):
            psf, dxy_psf = ref.generate_psf(**kw)
        out[f"{name}/psf"] = np.asarray(psf, np.float32)
        out[f"{name}/dxy_psf"] = np.array(dxy_psf, np.float64)
        out[f"{name}/kwargs"] = np.array(repr(sorted(kw.items())))
        print(name, psf.shape, float(psf.sum()), dxy_psf)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "psf_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
