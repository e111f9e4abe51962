"""Constants the hot path reads from the reference's ``utils.py:5-32`` (GPS helpers are out of scope)."""
import numpy as np

Camera_height = 1.65           # utils.py:6
SatMap_original_sidelength = 512
SatMap_process_sidelength = 512
Default_lat = 49.015
Satmap_zoom = 18
EPS = 1e-7


def get_camera_height():
    return Camera_height


def get_process_satmap_sidelength():
    return SatMap_process_sidelength


def get_meter_per_pixel(lat=Default_lat, zoom=Satmap_zoom, scale=SatMap_process_sidelength / SatMap_original_sidelength):
    """utils.py:28-32 -> 0.19582850865 m/px with the defaults."""
    m = 156543.03392 * np.cos(lat * np.pi / 180.0) / (2 ** zoom)
    m /= 2
    m /= scale
    return m
