"""fetal_t2mapping_amd -- MI355X-native per-voxel T2 relaxation fit.

Drop-in for the voxel-wise fitting path of Medical-Image-Analysis-Laboratory/fetal_t2mapping
(run_t2mapping.py fit_voxel / process_t2maps + utils/t2map_utils.compute_residuals).  The fit runs
in hand-written HIP kernels (csrc/) behind the C ABI of include/t2fit.h; this package is the
Python host side that mirrors the reference's function surface.  There is no CPU execution path.
"""
from .t2map import (T2Maps, compute_residuals, fit_table, fit_volume, fit_voxel, fit_voxels, fit_voxels_trace, label_stats,
                    make_config, set_fit_params, stack_mask_flatten, union_mask_dev)

__all__ = ["T2Maps", "compute_residuals", "fit_table", "fit_volume", "fit_voxel", "fit_voxels", "fit_voxels_trace", "label_stats", "make_config",
           "set_fit_params", "stack_mask_flatten", "union_mask_dev"]
