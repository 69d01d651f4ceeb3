"""`from model import G_NET, D_NET64, ...` as the reference's trainer.py:24 / interpolation.py do."""
from speech_to_image_translation_without_text_amd.model import *  # noqa: F401,F403
from speech_to_image_translation_without_text_amd.model import (CA_NET, D_NET64, D_NET128, D_NET256, D_NET512,  # noqa: F401
                                                                 D_NET1024, G_NET, GET_IMAGE_G, GLU, INCEPTION_V3,
                                                                 INIT_STAGE_G, NEXT_STAGE_G, ResBlock)
