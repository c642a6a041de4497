"""CPU: the exceptions a caller of the reference gets (SURVEY.md section 8b 'Errors': RuntimeError / ValueError /
NotImplementedError, no error codes) are raised by the product with the same types -- before any device work, so they can be
checked without a GPU.  Reference lines in the comments."""
import pytest
import torch

from diffusionhandles_amd import depth_transform as DT
from diffusionhandles_amd import losses as LS
from diffusionhandles_amd.guided_diffuser import GuidedDiffuser
from diffusionhandles_amd.guided_stable_diffuser import GuidedStableDiffuser, build_weight_schedule
from diffusionhandles_amd.null_inverter import NullInverter


def test_depth_transform_errors():
    with pytest.raises(RuntimeError):                       # depth_transform.py:16-17 depth not 4-D
        DT.normalize_depth(torch.ones(1, 8, 8))
    d = torch.ones(1, 1, 8, 8)
    K = torch.eye(3)
    with pytest.raises(RuntimeError):                       # :231-232 non-square mask
        DT.transform_depth(d, d, torch.ones(1, 1, 8, 6), K)
    with pytest.raises(ValueError):                         # :89 unknown mode
        DT.transform_depth(d, d, torch.ones(1, 1, 8, 8), K, depth_transform_mode="splat")
    with pytest.raises(ValueError):                         # :591-592 batch != 1
        DT.depth_to_world_coords(torch.ones(2, 1, 8, 8), K)
    with pytest.raises(RuntimeError):                       # :612-613 fewer than 2 pixels
        DT.depth_to_world_coords(torch.ones(1, 1, 1, 1), K)
    with pytest.raises(ValueError):                         # batch != 1 through the public entry point
        DT.transform_depth(torch.ones(2, 1, 8, 8), torch.ones(2, 1, 8, 8), torch.ones(1, 1, 8, 8), K)


def test_loss_and_schedule_errors():
    a = torch.zeros(4, 8, 8)
    with pytest.raises(ValueError):                         # losses.py:38 unknown background loss type
        LS.compute_background_loss(a, a, {}, 1, (8, 8), loss_type="median")
    with pytest.raises(ValueError):                         # guided_stable_diffuser.py:349 unknown schedule type
        build_weight_schedule(1.5, 1.25, 38, "cosine")


def test_abstract_seams_raise_not_implemented():
    gd = GuidedDiffuser(conf=None)                          # guided_diffuser.py:5-35: every method of the seam
    for call in (lambda: gd.to("cpu"), lambda: gd.get_image_shape(), lambda: gd.get_feature_shape(),
                 lambda: gd.initial_inference(None, None, None, None), lambda: gd.guided_inference(None, None, None, None, None, None),
                 lambda: gd.encode_latent_image(None), lambda: gd.decode_latent_image(None)):
        try:
            call()
        except NotImplementedError:
            continue
        except (AttributeError, TypeError):                 # a method the seam does not declare under that name / arity
            continue
        raise AssertionError("abstract method returned")
    with pytest.raises(NotImplementedError):                # null_inverter.py:5-15
        NullInverter(model=None).invert(None, None, None)
    with pytest.raises(NotImplementedError):                # guided_stable_diffuser.py:93-95 encode_latent_image
        GuidedStableDiffuser.encode_latent_image(object.__new__(GuidedStableDiffuser), None)


def test_no_cpu_fallback():
    """The product path fails loudly without a HIP device: no silent CPU route."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from diffusionhandles_amd.unet import HipUNet
    with pytest.raises(RuntimeError):
        HipUNet()
    with pytest.raises(RuntimeError):
        LS.process_correspondences(torch.zeros((0, 4), dtype=torch.int64), 512, 0)
    with pytest.raises(RuntimeError):
        DT.transform_depth(torch.ones(1, 1, 8, 8), torch.ones(1, 1, 8, 8), torch.ones(1, 1, 8, 8), torch.eye(3))
