"""GPU tier: the drop-in API in the reference's optimisation flow (core/luneburg_opt.py :33-128 +
core/optimizer.py :44-84): a coarse-to-fine Adam loop through tracer.BackTracerC must drive the
focusing loss down -- forward march, plane-intersection loss, autograd into the adjoint march and
the gradient mask/clamp all have to cooperate for that."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


@pytest.mark.gpu
def test_lens_design_loop_reduces_focus_loss(gpu):
    import luneburg_demo
    from adjointnonlinearraytracing_amd import drrt
    drrt.options.check_failed = False
    drrt.options.sort_rays = True
    n, hist = luneburg_demo.run(res_list=(9, 17), iters=40, pixels=48, verbose=False)
    first, last = sum(hist[:3]) / 3, sum(hist[-3:]) / 3
    assert last < 0.5 * first, (first, last)
    assert float(n.min()) >= 1.0 and float(n.max()) > 1.005         # a lens formed; clamp respected
    # the optimised medium is denser in the middle than at the rim (a focusing GRIN profile)
    c = n.shape[0] // 2
    assert float(n[c, c, c]) > float(n[c, c, 1])


@pytest.mark.gpu
def test_tomography_loop_recovers_the_field(gpu):
    """examples/tomography_demo.py: device ray generation -> march -> sensor images -> MSE -> adjoint -> Adam, with
    the multires hand-over of volume and Adam moments: the image loss and the reconstruction error must fall."""
    import tomography_demo
    from adjointnonlinearraytracing_amd import drrt
    drrt.options.check_failed = False
    drrt.options.sort_rays = True
    n, truth, hist, err = tomography_demo.run(res_list=(9, 17), views=4, iters=40, nbins=32, verbose=False)
    assert drrt.options.corrected_h is False                               # the demo restores the default
    first, last = sum(hist[:3]) / 3, sum(hist[-3:]) / 3
    assert last < 0.5 * first, (first, last)
    assert err[-1] < 0.8 * err[0], (err[0], err[-1])
    assert float(n.min()) >= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("sdf_weight", [0.0, 0.05])
def test_image_experiment_flow(gpu, sdf_weight):
    """examples/image_demo.py: the flow of core/image_opt.py on this package's mirrors -- area sources, near and
    far-field sensor images, sum_norm, the optional SDF-texture term, multires_opt with the fused Adam tail: the
    image loss must fall and the volume stay clamped.  (A design problem, not tomography: two views do not determine
    the volume, so the distance to the hidden volume is reported by the demo but not asserted.)"""
    import image_demo
    from adjointnonlinearraytracing_amd import drrt
    drrt.options.check_failed = False
    n, truth, hist, err = image_demo.run(res_list=(9, 17), views=2, iters=25, nbins=24, sdf_weight=sdf_weight,
                                         verbose=False)
    assert drrt.options.corrected_h is False
    assert n.shape == (17, 17, 17) and len(hist) == len(err) == 25 + 50
    first, last = sum(hist[:3]) / 3, sum(hist[-3:]) / 3
    # the images are Monte-Carlo estimates (16 samples per pixel): most of the MSE is their noise floor, the part the
    # volume can explain falls by about a tenth of the total here (seeded: 0.0303 -> 0.0274)
    assert np.isfinite(hist).all() and last < 0.96 * first, (first, last)
    assert float(n.min()) >= 1.0 and float(n.max()) > 1.005


@pytest.mark.gpu
@pytest.mark.parametrize("src", ["cone", "planar"])
def test_fibre_experiment_flow(gpu, src):
    """examples/fiber_demo.py: the flow of core/fiber_opt.py -- cone / plane source, boundary index through
    a radial index lookup, tracer.BackCableTracerC towards two targets a hop apart, Adam on the radial profile with the
    experiment's midpoint up-sampling: the refocusing loss must fall by a large factor."""
    import fiber_demo
    n, hist = fiber_demo.run(res_list=(5, 9), iters=20, nbins=24, src_type=src, verbose=False)
    assert n.shape == (9,) and len(hist) == 20 + 40 and np.isfinite(hist).all()
    first, last = sum(hist[:3]) / 3, sum(hist[-3:]) / 3
    assert last < 0.5 * first, (first, last)
    assert abs(float(n[-1]) - 1.0) < 1e-6                                # the cladding sample is never updated (:240)
