"""Drop-in module path of the reference (`from src.solver import Trainer`); the implementation lives in the
MI355X-native package `end-to-end-asr-pytorch_amd/`."""
