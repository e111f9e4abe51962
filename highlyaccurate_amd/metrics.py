"""Evaluation metrics of the reference's test loops (``train_kitti.py:77-170`` test1/test2, ``train_ford.py:77-180``):
distance / lateral / longitudinal / heading recall at {1, 3, 5} (metres, degrees) for the predictions and for the
initial pose, with the same text lines the reference prints and appends to ``Test*_results.txt``.  Plain numpy."""
from __future__ import annotations

import numpy as np

METRICS = (1, 3, 5)     # metres
ANGLES = (1, 3, 5)      # degrees


def localisation_metrics(pred_shifts, pred_headings, gt_shifts, gt_headings, shift_range_lat, shift_range_lon,
                         rotation_range, reference_compat: bool = False):
    """All inputs in the model's normalised units: pred_shifts / gt_shifts [N,2] = (lat, lon), headings [N,1].
    Returns (result, stats dict, text lines).  ``result`` is the model-selection score (train_kitti.py:160): the percentage
    of samples within 1 m AND 1 degree.  The reference computes it as ``(distance < 1) & (angle_diff < 1)`` with distance
    [N] and angle_diff [N,1], which BROADCASTS to an N x N mask (every distance paired with every angle; the score can
    exceed 100).  The default here is the per-sample score the line means; ``reference_compat=True`` reproduces the
    reference's number bit for bit, for anyone who must select the same best checkpoint as an existing run."""
    scale = np.array([shift_range_lat, shift_range_lon], dtype=np.float64).reshape(1, 2)
    # dtypes as in the reference (77-80): shifts are multiplied by a float64 ARRAY (-> float64), headings by a python
    # float (-> they stay in the network's float32), and every later comparison is made on those values
    ps, gs = np.asarray(pred_shifts) * scale, np.asarray(gt_shifts) * scale
    ph, gh = np.asarray(pred_headings) * rotation_range, np.asarray(gt_headings) * rotation_range
    distance = np.sqrt(np.sum((ps - gs) ** 2, axis=1))
    angle_diff = np.remainder(np.abs(ph - gh), 360)
    angle_diff = np.where(angle_diff > 180, 360 - angle_diff, angle_diff)
    init_dis = np.sqrt(np.sum(gs ** 2, axis=1))
    init_angle = np.abs(gh)
    diff = np.abs(ps - gs)
    n = float(distance.shape[0])
    stats = {'init_distance_mean': float(np.mean(init_dis)), 'pred_distance_mean': float(np.mean(distance)),
             'init_angle_mean': float(np.mean(init_angle)), 'pred_angle_mean': float(np.mean(angle_diff))}
    lines = []
    for m in METRICS:
        stats[f'distance@{m}'] = (np.sum(distance < m) / n * 100, np.sum(init_dis < m) / n * 100)
        lines.append(f'distance within {m} meters (pred, init): {stats[f"distance@{m}"][0]} {stats[f"distance@{m}"][1]}')
    lines.append('------------------------')
    for m in METRICS:
        stats[f'lateral@{m}'] = (np.sum(diff[:, 0] < m) / n * 100, np.sum(np.abs(gs[:, 0]) < m) / n * 100)
        stats[f'longitudinal@{m}'] = (np.sum(diff[:, 1] < m) / n * 100, np.sum(np.abs(gs[:, 1]) < m) / n * 100)
        lines.append(f'lateral      within {m} meters (pred, init): {stats[f"lateral@{m}"][0]} {stats[f"lateral@{m}"][1]}')
        lines.append(f'longitudinal within {m} meters (pred, init): {stats[f"longitudinal@{m}"][0]} {stats[f"longitudinal@{m}"][1]}')
    lines.append('------------------------')
    for a in ANGLES:
        stats[f'angle@{a}'] = (np.sum(angle_diff < a) / n * 100, np.sum(init_angle < a) / n * 100)
        lines.append(f'angle within {a} degrees (pred, init): {stats[f"angle@{a}"][0]} {stats[f"angle@{a}"][1]}')
    lines.append('------------------------')
    for m, a in zip(METRICS, ANGLES):
        p = np.sum((angle_diff[:, 0] < a) & (diff[:, 0] < m)) / n * 100
        i = np.sum((init_angle[:, 0] < a) & (np.abs(gs[:, 0]) < m)) / n * 100
        stats[f'lat@{m}&angle@{a}'] = (p, i)
        lines.append(f'lat within {m} & angle within {a} (pred, init): {p} {i}')
    # both scores are always reported, so that checkpoint selection can be compared with an existing run either way
    # the reference's expression broadcasts [N] & [N,1] to an [N,N] matrix and sums it (train_kitti.py:163-164): that sum is
    # count(distance < 1 m) * count(angle < 1 deg) -- computed as such, without the N^2 temporaries
    stats['result_reference'] = float(np.sum(distance < METRICS[0]) * np.sum(angle_diff < ANGLES[0]) / n * 100)
    stats['result_per_sample'] = float(np.sum((distance < METRICS[0]) & (angle_diff[:, 0] < ANGLES[0])) / n * 100)
    result = stats['result_reference'] if reference_compat else stats['result_per_sample']
    return result, stats, lines


def write_test_results(save_path, name, epoch, duration, pred_shifts, pred_headings, gt_shifts, gt_headings,
                       shift_range_lat, shift_range_lon, rotation_range, reference_compat: bool = False):
    """What test1 / test2 leave on disk (train_kitti.py:77-161, 212-296; train_ford.py the same):
      ``<save_path>/<name>_results.mat``  gt_shifts, gt_headings, pred_shifts, pred_headings in metres / degrees (line 82)
      ``<save_path>/<name>_results.txt``  one block per call APPENDED: header, EPOCH, time per image, the recall lines
    ``name`` is 'Test1' or 'Test2'.  Inputs in the model's normalised units.  Returns (result, stats)."""
    import os
    import scipy.io as scio
    os.makedirs(save_path, exist_ok=True)
    scale = np.array([shift_range_lat, shift_range_lon], dtype=np.float64).reshape(1, 2)
    scio.savemat(os.path.join(save_path, f'{name}_results.mat'),
                 {'gt_shifts': np.asarray(gt_shifts) * scale, 'gt_headings': np.asarray(gt_headings) * rotation_range,
                  'pred_shifts': np.asarray(pred_shifts) * scale, 'pred_headings': np.asarray(pred_headings) * rotation_range})
    result, stats, lines = localisation_metrics(pred_shifts, pred_headings, gt_shifts, gt_headings, shift_range_lat,
                                                shift_range_lon, rotation_range, reference_compat)
    with open(os.path.join(save_path, f'{name}_results.txt'), 'a') as f:
        f.write('====================================\n')
        f.write('       EPOCH: ' + str(epoch) + '\n')
        f.write('Time per image (second): ' + str(duration) + '\n')
        for line in lines:
            f.write(line + '\n')
        f.write('====================================\n')
    return result, stats
