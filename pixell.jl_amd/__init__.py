"""pixell.jl_amd -- MI355X-native (gfx950) implementation of the Pixell.jl CAR pixel<->sky hot path.

Host-side mirror of the reference's Enmap/WCS interface (same names, argument meaning and `safe`
behaviour as /root/reference/src/Pixell.jl:35-43 exports for this path) over libpixell_hip.so, a C-ABI
library of hand-written HIP kernels (include/pixell_hip.h).  PyTorch only supplies device memory,
streams and torch.distributed.  The directory name contains a dot, so import it through the
repo-root shim:  `import pixell_jl_amd as pj`.
"""
from . import _lib
from ._lib import PixellHipError, LIB_PATH
from .wcs import (AbstractCARWCS, CarClenshawCurtis, CarFejer1, Gnomonic, SkyBoundingBox, getcdelt, getcrpix,
                  getcrval, getunit, is_periodic, jl_mod, rewind, sliced_wcs)
from .geometry import (JlRange, create_car_wcs, extent_cyl, fullsky_geometry, geometry, laxes_cyl, pad_geometry,
                       skyarea, slice_geometry)
from .enmap import Enmap, NoWCS, getwcs
from .ops import (GenericReprojectPlan, ReprojectPlan, SamplePairs, fill_random_, fill_sphere_points_, pix2sky, pix2sky_, pix2sky_rewind,
                  pixareamap, pixareamap_, posmap, reproject, rewind_, sample_bilinear, sky2pix, sky2pix_,
                  sky2pix_broadcast, unwind_)
from .sharding import DecStripLayout, DecStripReprojector, strip_bounds
from .placement import (allocation_policy, empty_map, last_allocation_info, map_classes, place_pair, place_pair_compact, place_pair_native, place_pair_shifted, place_streams,
                        set_allocation_policy)
from .fits_io import read_header, read_map, read_map_rows, wcs_from_header, write_map

# unit shortcuts, Pixell.jl:46-48 (angles are plain radians here)
import math as _math
radian = 1.0
degree = _math.pi / 180
arcminute = _math.pi / 180 / 60


def library_path():
    return LIB_PATH


def load_library():
    """Load libpixell_hip.so now (raises ImportError if it has not been built)."""
    return _lib.load()
