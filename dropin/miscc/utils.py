from speech_to_image_translation_without_text_amd.miscc.utils import mkdir_p  # noqa: F401
