"""mkdir_p, as imported by the reference trainer (StackGAN_v2/trainer.py:18)."""
import os


def mkdir_p(path):
    os.makedirs(path, exist_ok=True)
