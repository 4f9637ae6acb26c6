"""Helpers with the names and argument meaning of python/src/fenicsxfus/utils.py."""
import numpy as np


def compute_diffusivity_of_sound(frequency: float, speed: float, attenuationdB: float) -> float:
    """Diffusivity of sound from an attenuation in dB/m (python/src/fenicsxfus/utils.py:50-55;
    ``frequency`` is the angular frequency the reference's callers pass)."""
    attenuationNp = attenuationdB / 20 * np.log(10)
    return 2 * attenuationNp * speed**3 / frequency**2
