"""`from trainer import condGANTrainer` as the reference's main.py:187 does."""
from speech_to_image_translation_without_text_amd.trainer import *  # noqa: F401,F403
from speech_to_image_translation_without_text_amd.trainer import (KL_loss, class_aware_loss, condGANTrainer,  # noqa: F401
                                                                   copy_G_params, define_optimizers, load_network,
                                                                   load_params, save_model, weights_init)
