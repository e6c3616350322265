"""A synthetic unpacked Neurofinder challenge directory (TIFF frames + regions.json): bright discs on noise."""
import json
import os

import numpy as np


def make_neurofinder_dir(datasets_dir, name, hw=(512, 512), neurons=60, frames=6, seed=11):
    from PIL import Image
    root = '%s/%s' % (datasets_dir, name)
    os.makedirs(root + '/images')
    rs = np.random.RandomState(seed)
    regions, base = [], rs.randint(200, 400, size=hw)
    for _ in range(neurons):
        cy, cx = rs.randint(8, hw[0] - 8), rs.randint(8, hw[1] - 8)
        coords = [[int(cy + dy), int(cx + dx)] for dy in range(-3, 4) for dx in range(-3, 4) if dy * dy + dx * dx <= 10]
        regions.append({'coordinates': coords})
        for y, x in coords:
            base[y, x] += 600
    for i in range(frames):
        Image.fromarray((base + rs.randint(0, 60, size=hw)).astype(np.uint16)).save('%s/images/image%05d.tiff' % (root, i))
    if '.test' not in name:
        os.makedirs(root + '/regions')
        with open(root + '/regions/regions.json', 'w') as fp:
            json.dump(regions, fp)
    return root
