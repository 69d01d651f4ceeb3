"""`from miscc.config import cfg, cfg_from_file` (reference main.py:21, trainer.py:17): the SAME cfg object
the package's model/trainer read."""
from speech_to_image_translation_without_text_amd.miscc.config import (cfg, cfg_from_dict, cfg_from_file,  # noqa: F401
                                                                        cfg_reset, edict)
