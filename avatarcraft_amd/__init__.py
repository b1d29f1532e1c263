"""avatarcraft_amd -- MI355X-native hot path of AvatarCraft (hash-grid NeuS renderer + SDS step).

Sub-packages mirror the reference's module surface so that its drivers are drop-in:
    avatarcraft_amd.encoder       <-> reference `encoder`      (get_encoder, HashEncoder, SHEncoder, freq_encoder)
    avatarcraft_amd.raymarching   <-> reference `raymarching`  (march_rays_train, composite_rays_train, ...)
    avatarcraft_amd.instant_nsr   <-> reference `models/instant_nsr.py` (NeRFNetwork.render)
    avatarcraft_amd.render_utils  <-> reference `utils/render_utils.py` (render_instantnsr_naive, ...)
`avatarcraft_amd.dropin.install()` registers these under the reference's top-level names.
All numerical work is done by libavatarcraft_hip.so (hand-written HIP for gfx950, C ABI in
include/avatarcraft_hip.h); importing works without a GPU, calling the ops does not.
"""
__version__ = "0.1.0"
