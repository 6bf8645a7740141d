"""deepatlas_amd -- MI355X (gfx950) native hot path of uncbiag/DeepAtlas.

Layout mirrors the reference's module surfaces:
  deepatlas_amd.lib.network_factory  get_network / get_available_networks ('UNet_light', 'UNet', 'voxel_morph_cvpr')
  deepatlas_amd.lib.loss             get_loss_function / get_available_losses ('dice', 'ncc', 'bendingEnergy', ...)
  deepatlas_amd.lib.transforms       mask_to_one_hot, SegMaskToOneHot, CropTensor, SitkToTensor (tensor level)
  deepatlas_amd.lib.utils            get_identity_transform(_batch), initialize_model
  deepatlas_amd.lib.evalMetrics      eval Dice on device
  deepatlas_amd.models.segmentation  SegmentationExperiment
  deepatlas_amd.csrc                 HIP kernels + the C ABI (include/deepatlas_hip.h)
All compute runs in libdeepatlas_hip.so; there is no CPU fallback (oracle/ is test infrastructure).
"""
__version__ = '0.1.0'
