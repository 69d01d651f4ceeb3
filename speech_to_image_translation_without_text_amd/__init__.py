"""MI355X-native StackGAN-v2 G/D train step (speech-to-image): HIP kernels behind the reference's
G_NET / D_NET* module surface.  See DESIGN.md."""
__all__ = ["_lib", "ops"]
