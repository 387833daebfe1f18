"""Height-field terrain: the sub-terrain generators and the curriculum grid.

Two layers, as in the reference:
  * generators (`SubTerrain`, `wave_terrain`, `random_uniform_terrain`, `pyramid_sloped_terrain`, `pyramid_stairs_terrain`,
    `discrete_obstacles_terrain`, `stepping_stones_terrain`) — in the reference these come from `isaacgym.terrain_utils`, a
    third-party module that is NOT part of /root/reference.  They are written here from their documented
    behaviour (int16 height units of `vertical_scale`, pixels of `horizontal_scale`); pixel-exact agreement with Isaac Gym's
    own implementation is unpinned.
  * `Terrain` — restates legged_gym/utils/terrain.py:9-197 (grid layout :24-36, curriculum :61-71, per-kind parameters :87-155,
    map placement and env origins :157-174, `gap_terrain` :176, `pit_terrain` :190).  Pinned: oracle/gen_golden.py runs the
    reference's Terrain on these same generators with the same numpy seed and tests/test_terrain.py compares every output.
One-off CPU initialisation (numpy); the per-step terrain work (contact queries, 187-point height scan) is in the kernels.
"""
from collections import defaultdict

import numpy as np
from scipy import interpolate


class SubTerrain:
    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale, self.horizontal_scale = vertical_scale, horizontal_scale
        self.width, self.length = width, length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def random_uniform_terrain(terrain, min_height, max_height, step=1, downsampled_scale=None):
    """Uniform noise on a coarse grid (spacing `downsampled_scale` m, heights quantised to `step` m), bilinearly upsampled."""
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    lo, hi, st = int(min_height / terrain.vertical_scale), int(max_height / terrain.vertical_scale), int(step / terrain.vertical_scale)
    heights_range = np.arange(lo, hi + st, st)
    nx, ny = int(terrain.width * terrain.horizontal_scale / downsampled_scale), int(terrain.length * terrain.horizontal_scale / downsampled_scale)
    coarse = np.random.choice(heights_range, (nx, ny))
    x, y = np.linspace(0, terrain.width * terrain.horizontal_scale, nx), np.linspace(0, terrain.length * terrain.horizontal_scale, ny)
    f = interpolate.RectBivariateSpline(x, y, coarse, kx=1, ky=1)
    xu, yu = np.linspace(0, terrain.width * terrain.horizontal_scale, terrain.width), np.linspace(0, terrain.length * terrain.horizontal_scale, terrain.length)
    terrain.height_field_raw += np.rint(f(xu, yu)).astype(np.int16)
    return terrain


def pyramid_sloped_terrain(terrain, slope=1, platform_size=1.0):
    """Pyramid with faces of the given slope (negative: inverted) and a flat square platform of `platform_size` m on top."""
    x, y = np.arange(0, terrain.width), np.arange(0, terrain.length)
    cx, cy = int(terrain.width / 2), int(terrain.length / 2)
    xx, yy = np.meshgrid((cx - np.abs(cx - x)) / cx, (cy - np.abs(cy - y)) / cy, sparse=True)
    xx, yy = xx.reshape(terrain.width, 1), yy.reshape(1, terrain.length)
    max_height = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (terrain.width / 2))
    terrain.height_field_raw += (max_height * xx * yy).astype(terrain.height_field_raw.dtype)
    ps = int(platform_size / terrain.horizontal_scale / 2)
    x1, x2, y1, y2 = terrain.width // 2 - ps, terrain.width // 2 + ps, terrain.length // 2 - ps, terrain.length // 2 + ps
    lo, hi = min(terrain.height_field_raw[x1, y1], 0), max(terrain.height_field_raw[x1, y1], 0)
    terrain.height_field_raw = np.clip(terrain.height_field_raw, lo, hi)
    return terrain


def pyramid_stairs_terrain(terrain, step_width, step_height, platform_size=1.0):
    """Concentric square steps (`step_width` m treads, `step_height` m risers; negative = going up towards the centre...
    sign as in the caller: terrain.py:121-127 passes -h for 'stairs up') with a central platform."""
    sw, sh, ps = int(step_width / terrain.horizontal_scale), int(step_height / terrain.vertical_scale), int(platform_size / terrain.horizontal_scale)
    height, x0, x1, y0, y1 = 0, 0, terrain.width, 0, terrain.length
    while (x1 - x0) > ps and (y1 - y0) > ps:
        x0 += sw; x1 -= sw; y0 += sw; y1 -= sw
        height += sh
        terrain.height_field_raw[x0:x1, y0:y1] = height
    return terrain


def discrete_obstacles_terrain(terrain, max_height, min_size, max_size, num_rects, platform_size=1.0):
    """`num_rects` random axis-aligned boxes of height in {-h, -h/2, h/2, h}, and a flat central platform."""
    mh, mn, mx, ps = int(max_height / terrain.vertical_scale), int(min_size / terrain.horizontal_scale), int(max_size / terrain.horizontal_scale), int(platform_size / terrain.horizontal_scale)
    (i, j) = terrain.height_field_raw.shape
    height_range = [-mh, -mh // 2, mh // 2, mh]
    size_range = range(mn, mx, 4)
    for _ in range(num_rects):
        w, l = np.random.choice(size_range), np.random.choice(size_range)
        si, sj = np.random.choice(range(0, i - w, 4)), np.random.choice(range(0, j - l, 4))
        terrain.height_field_raw[si:si + w, sj:sj + l] = np.random.choice(height_range)
    x1, x2, y1, y2 = (terrain.width - ps) // 2, (terrain.width + ps) // 2, (terrain.length - ps) // 2, (terrain.length + ps) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def wave_terrain(terrain, num_waves=1, amplitude=1.0):
    """Sum of a cosine along y and a sine along x, `num_waves` periods over the tile, peak-to-peak `amplitude` m each."""
    amp = int(0.5 * amplitude / terrain.vertical_scale)
    if num_waves > 0:
        div = terrain.length / (num_waves * np.pi * 2)
        xx, yy = np.meshgrid(np.arange(0, terrain.width), np.arange(0, terrain.length), sparse=True)
        xx, yy = xx.reshape(terrain.width, 1), yy.reshape(1, terrain.length)
        terrain.height_field_raw += (amp * np.cos(yy / div) + amp * np.sin(xx / div)).astype(terrain.height_field_raw.dtype)
    return terrain


def stepping_stones_terrain(terrain, stone_size, stone_distance, max_height, platform_size=1.0, depth=-10):
    """Square stones of `stone_size` m separated by `stone_distance` m over a pit `depth` m deep, random stone heights."""
    ss, sd, mh, ps = int(stone_size / terrain.horizontal_scale), int(stone_distance / terrain.horizontal_scale), int(max_height / terrain.vertical_scale), int(platform_size / terrain.horizontal_scale)
    height_range = np.arange(-mh - 1, mh, step=1)
    sx, sy = 0, 0
    terrain.height_field_raw[:, :] = int(depth / terrain.vertical_scale)
    if terrain.length >= terrain.width:
        while sy < terrain.length:
            ey = min(terrain.length, sy + ss)
            sx = np.random.randint(0, ss)
            ex = max(0, sx - sd)
            terrain.height_field_raw[0:ex, sy:ey] = np.random.choice(height_range)
            while sx < terrain.width:
                ex = min(terrain.width, sx + ss)
                terrain.height_field_raw[sx:ex, sy:ey] = np.random.choice(height_range)
                sx += ss + sd
            sy += ss + sd
    else:
        while sx < terrain.width:
            ex = min(terrain.width, sx + ss)
            sy = np.random.randint(0, ss)
            ey = max(0, sy - sd)
            terrain.height_field_raw[sx:ex, 0:ey] = np.random.choice(height_range)
            while sy < terrain.length:
                ey = min(terrain.length, sy + ss)
                terrain.height_field_raw[sx:ex, sy:ey] = np.random.choice(height_range)
                sy += ss + sd
            sx += ss + sd
    x1, x2, y1, y2 = (terrain.width - ps) // 2, (terrain.width + ps) // 2, (terrain.length - ps) // 2, (terrain.length + ps) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def gap_terrain(terrain, gap_size, platform_size=1.0):             # legged_gym/utils/terrain.py:176-188
    gap, ps = int(gap_size / terrain.horizontal_scale), int(platform_size / terrain.horizontal_scale)
    cx, cy = terrain.length // 2, terrain.width // 2
    x1 = (terrain.length - ps) // 2; x2 = x1 + gap
    y1 = (terrain.width - ps) // 2; y2 = y1 + gap
    terrain.height_field_raw[cx - x2:cx + x2, cy - y2:cy + y2] = -1000
    terrain.height_field_raw[cx - x1:cx + x1, cy - y1:cy + y1] = 0


def pit_terrain(terrain, depth, platform_size=1.0):                 # legged_gym/utils/terrain.py:190-197
    d, ps = int(depth / terrain.vertical_scale), int(platform_size / terrain.horizontal_scale / 2)
    x1, x2 = terrain.length // 2 - ps, terrain.length // 2 + ps
    y1, y2 = terrain.width // 2 - ps, terrain.width // 2 + ps
    terrain.height_field_raw[x1:x2, y1:y2] = -d


def convert_heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None):
    """The triangle mesh PhysX gets for mesh_type 'trimesh' (isaacgym.terrain_utils, third-party and absent; restated from its documented
    behaviour; used at legged_gym/utils/terrain.py:46-49).  One vertex per sample at (i*hs, j*hs, h*vs), two triangles per cell split along
    (i,j)-(i+1,j+1).  With a slope threshold, the LOWER vertex of every edge steeper than it is moved one cell towards the higher one
    (axis moves first, the diagonal move only where no axis move happened), which turns 1-cell ramps into vertical walls.
    -> vertices float32 [rows*cols, 3], triangles uint32 [2*(rows-1)*(cols-1), 3]."""
    hf = np.asarray(height_field_raw)
    rows, cols = hf.shape
    yy, xx = np.meshgrid(np.linspace(0, (cols - 1) * horizontal_scale, cols), np.linspace(0, (rows - 1) * horizontal_scale, rows))
    if slope_threshold is not None:
        thr = slope_threshold * horizontal_scale / vertical_scale           # in height units per cell
        mvx, mvy = _vertex_moves(hf.astype(np.int64), thr)
        xx = xx + mvx * horizontal_scale
        yy = yy + mvy * horizontal_scale
    vertices = np.zeros((rows * cols, 3), dtype=np.float32)
    vertices[:, 0], vertices[:, 1], vertices[:, 2] = xx.flatten(), yy.flatten(), hf.flatten() * vertical_scale
    triangles = -np.ones((2 * (rows - 1) * (cols - 1), 3), dtype=np.uint32)
    for i in range(rows - 1):
        ind0 = np.arange(0, cols - 1) + i * cols
        ind1, ind2, ind3 = ind0 + 1, ind0 + cols, ind0 + cols + 1
        a, b = 2 * i * (cols - 1), 2 * (i + 1) * (cols - 1)
        triangles[a:b:2, 0], triangles[a:b:2, 1], triangles[a:b:2, 2] = ind0, ind3, ind1
        triangles[a + 1:b:2, 0], triangles[a + 1:b:2, 1], triangles[a + 1:b:2, 2] = ind0, ind2, ind3
    return vertices, triangles


def _vertex_moves(h, thr):
    """Per-vertex displacement (in cells) of convert_heightfield_to_trimesh's slope correction -> (move_x, move_y) int arrays in {-1, 0, 1}."""
    rows, cols = h.shape
    move_x, move_y, move_c = np.zeros((rows, cols)), np.zeros((rows, cols)), np.zeros((rows, cols))
    move_x[:rows - 1, :] += h[1:, :] - h[:rows - 1, :] > thr
    move_x[1:, :] -= h[:rows - 1, :] - h[1:, :] > thr
    move_y[:, :cols - 1] += h[:, 1:] - h[:, :cols - 1] > thr
    move_y[:, 1:] -= h[:, :cols - 1] - h[:, 1:] > thr
    move_c[:rows - 1, :cols - 1] += h[1:, 1:] - h[:rows - 1, :cols - 1] > thr
    move_c[1:, 1:] -= h[:rows - 1, :cols - 1] - h[1:, 1:] > thr
    return (move_x + move_c * (move_x == 0)).astype(np.int64), (move_y + move_c * (move_y == 0)).astype(np.int64)


def displaced_cell_heights(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None):
    """The surface of the trimesh the reference hands PhysX (convert_heightfield_to_trimesh WITH its slope_treshold vertex displacement,
    legged_gym/utils/terrain.py:46-49, legged_robot.py:1127-1141) in the form the simulator's contact query reads: for every grid cell
    (i, j) the heights [h00, h10, h01, h11] of the displaced surface at the cell's four corners AS SEEN FROM INSIDE THE CELL
    (int16, vertical_scale units) -> int16 [rows-1, cols-1, 4].

    Without displacement the four numbers are the shared vertex samples and neighbouring cells agree on their common edge.  Where the
    correction moved the lower vertex of a steep edge under the upper one, the 1-cell ramp becomes a flat continuation of the lower level
    ending in a vertical face: neighbouring cells then DISAGREE on their common edge, and that disagreement is the wall (its bottom = this
    cell's edge heights, its top = the neighbour's).  Exact wherever the displaced surface over a cell is one plane or two planes split along
    the cell's own diagonal (stairs, obstacles, stepping stones, gaps, pits, gentle slopes); elsewhere the cell's two facets interpolate
    the displaced surface at the corners.  Only cells within two cells of a displaced vertex are recomputed (geometrically: the displaced
    triangle covering a point just inside each corner, highest one if the displacement folded the surface)."""
    hf = np.asarray(height_field_raw)
    h = hf.astype(np.int64)
    rows, cols = h.shape
    cells = np.stack([h[:-1, :-1], h[1:, :-1], h[:-1, 1:], h[1:, 1:]], axis=-1).astype(np.float64)
    if slope_threshold is None:
        return cells.astype(np.int16)
    mx, my = _vertex_moves(h, slope_threshold * horizontal_scale / vertical_scale)
    moved = (mx != 0) | (my != 0)
    if not moved.any():
        return cells.astype(np.int16)
    pad = np.pad(moved, 2)
    aff = np.zeros((rows - 1, cols - 1), bool)
    for a in range(4):
        for b in range(4):
            aff |= pad[1 + a:1 + a + rows - 1, 1 + b:1 + b + cols - 1]
    ii, jj = np.nonzero(aff)
    vx, vy, vz = np.arange(rows)[:, None] + mx, np.arange(cols)[None, :] + my, h.astype(np.float64)
    eps = 1e-3
    for k, (cx, cy) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        px = ii + cx + (eps if cx == 0 else -eps)
        py = jj + cy + (eps if cy == 0 else -eps)
        best = np.full(ii.shape, -np.inf)
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                a_, b_ = ii - di, jj - dj
                ok = (a_ >= 0) & (a_ < rows - 1) & (b_ >= 0) & (b_ < cols - 1)
                a, b = np.clip(a_, 0, rows - 2), np.clip(b_, 0, cols - 2)
                for tri in (((0, 0), (1, 1), (0, 1)), ((0, 0), (1, 0), (1, 1))):      # triangles (ind0, ind3, ind1), (ind0, ind2, ind3)
                    X = [vx[a + o[0], b + o[1]].astype(np.float64) for o in tri]
                    Y = [vy[a + o[0], b + o[1]].astype(np.float64) for o in tri]
                    Z = [vz[a + o[0], b + o[1]] for o in tri]
                    det = (Y[1] - Y[2]) * (X[0] - X[2]) + (X[2] - X[1]) * (Y[0] - Y[2])
                    good = ok & (np.abs(det) > 1e-9)
                    det = np.where(good, det, 1.0)
                    l0 = ((Y[1] - Y[2]) * (px - X[2]) + (X[2] - X[1]) * (py - Y[2])) / det
                    l1 = ((Y[2] - Y[0]) * (px - X[2]) + (X[0] - X[2]) * (py - Y[2])) / det
                    l2 = 1.0 - l0 - l1
                    inside = good & (l0 >= -1e-9) & (l1 >= -1e-9) & (l2 >= -1e-9)
                    z = l0 * Z[0] + l1 * Z[1] + l2 * Z[2]
                    best = np.where(inside & (z > best), z, best)
        found = np.isfinite(best)
        cells[ii[found], jj[found], k] = best[found]
    return np.clip(np.rint(cells), -32768, 32767).astype(np.int16)


KIND_NAMES = ("wave", "slope", "rough_slope", "stairs_up", "stairs_down", "obstacles", "stepping_stones", "gap", "flat")


class Terrain:
    def __init__(self, cfg, num_robots):
        self.cfg, self.num_robots, self.type = cfg, num_robots, cfg.mesh_type
        if self.type in ("none", "plane"):
            return
        self.env_length, self.env_width = cfg.terrain_length, cfg.terrain_width
        self.proportions = [np.sum(cfg.terrain_proportions[:i + 1]) for i in range(len(cfg.terrain_proportions))]
        self.cfg.num_sub_terrains = cfg.num_rows * cfg.num_cols
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        hs = cfg.horizontal_scale
        self.width_per_env_pixels, self.length_per_env_pixels = int(self.env_width / hs), int(self.env_length / hs)
        self.spacing = cfg.terrain_spacing
        self.spacing_pixels = int(self.spacing / hs)
        self.border = int(cfg.border_size / hs)
        self.tot_cols = int(cfg.num_cols * self.width_per_env_pixels + max(0, cfg.num_cols - 1) * self.spacing_pixels) + 2 * self.border
        self.tot_rows = int(cfg.num_rows * self.length_per_env_pixels + max(0, cfg.num_rows - 1) * self.spacing_pixels) + 2 * self.border
        self.name2cols = defaultdict(set)
        self.cols2id = []
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        if cfg.curriculum:
            self.curiculum()
        elif cfg.selected:
            self.selected_terrain()
        else:
            self.randomized_terrain()
        self.heightsamples = self.height_field_raw
        # mesh_type 'trimesh' (the reference's default) converts this same height field to triangles for PhysX (:45-49); this build's
        # contact queries work on the height field directly, so the mesh is only built when somebody asks for it (vertices / triangles).
        self._trimesh = None

    def _mesh(self):
        if self._trimesh is None:
            self._trimesh = convert_heightfield_to_trimesh(self.height_field_raw, self.cfg.horizontal_scale, self.cfg.vertical_scale, self.cfg.slope_treshold)
        return self._trimesh

    @property
    def vertices(self):
        return self._mesh()[0]

    @property
    def triangles(self):
        return self._mesh()[1]

    @property
    def cell_heights(self):
        """int16 [tot_rows-1, tot_cols-1, 4]: what the simulator's contact query reads for mesh_type 'trimesh' — the displaced mesh's surface,
        cell by cell (displaced_cell_heights).  For 'heightfield' the library derives the (continuous) cells from the samples itself."""
        if getattr(self, "_cells", None) is None:
            thr = self.cfg.slope_treshold if self.type == "trimesh" else None
            self._cells = displaced_cell_heights(self.height_field_raw, self.cfg.horizontal_scale, self.cfg.vertical_scale, thr)
        return self._cells

    def selected_terrain(self):
        """Every tile from ONE named generator (terrain.py:72-85; the reference's version dereferences attributes that do not
        exist and cannot run — this is what it evidently means).  Generators are looked up by name, never eval'd."""
        kwargs = dict(self.cfg.terrain_kwargs)
        gen = _GENERATORS[kwargs.pop("type")]
        for k in range(self.cfg.num_sub_terrains):
            (i, j) = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            t = SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                           vertical_scale=self.cfg.vertical_scale, horizontal_scale=self.cfg.horizontal_scale)
            gen(t, **kwargs)
            self.add_terrain_to_map(t, i, j)

    def randomized_terrain(self):
        for k in range(self.cfg.num_sub_terrains):
            (i, j) = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            choice = np.random.uniform(0, 1)
            difficulty = np.random.choice([0.5, 0.75, 0.9])
            self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)

    def curiculum(self):
        for j in range(self.cfg.num_cols):
            for i in range(self.cfg.num_rows):
                terrain = self.make_terrain(j / self.cfg.num_cols + 0.001, i / self.cfg.num_rows)
                self.add_terrain_to_map(terrain, i, j)
            self.name2cols[terrain.terrain_name].add(j)
            self.cols2id.append(terrain.terrain_id)

    def make_terrain(self, choice, difficulty):
        t = SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                       vertical_scale=self.cfg.vertical_scale, horizontal_scale=self.cfg.horizontal_scale)
        slope = 0.1 + difficulty * 0.52                 # the reference's "hard" parameter set (:93-98)
        step_height = 0.05 + 0.23 * difficulty
        obstacles_height = 0.05 + difficulty * 0.25
        stones_size = 1.5 * (1.05 - difficulty)
        stone_distance = 0.05 if difficulty == 0 else 0.1
        gap_size = 1.0 * difficulty
        amplitude = 0.1 + 0.2 * difficulty
        p = self.proportions

        def kind(k):
            t.terrain_name, t.terrain_id = KIND_NAMES[k], k
        if choice < p[0]:
            kind(0); wave_terrain(t, num_waves=5, amplitude=amplitude)
            random_uniform_terrain(t, min_height=-0.05, max_height=0.05, step=0.005, downsampled_scale=0.2)
        elif choice < p[1]:
            kind(1)
            if choice < (p[0] + p[1]) / 2:
                slope *= -1
            pyramid_sloped_terrain(t, slope=slope, platform_size=3.0)
        elif choice < p[2]:
            kind(2); pyramid_sloped_terrain(t, slope=slope, platform_size=3.0)
            random_uniform_terrain(t, min_height=-0.05, max_height=0.05, step=0.005, downsampled_scale=0.2)
        elif choice < p[4]:
            kind(4)
            if choice < p[3]:
                kind(3); step_height *= -1
            pyramid_stairs_terrain(t, step_width=0.31, step_height=step_height, platform_size=3.0)
        elif choice < p[5]:
            kind(5); discrete_obstacles_terrain(t, obstacles_height, 1.0, 2.0, 20, platform_size=3.0)
        elif choice < p[6]:
            kind(6); stepping_stones_terrain(t, stone_size=stones_size, stone_distance=stone_distance, max_height=0.0, platform_size=4.0)
        elif choice < p[7]:
            kind(7); gap_terrain(t, gap_size=gap_size, platform_size=3.0)
        else:
            kind(8); pit_terrain(t, depth=0.0, platform_size=4.0)
        return t

    def add_terrain_to_map(self, terrain, row, col):
        i, j = row, col
        sx = self.border + i * (self.length_per_env_pixels + self.spacing_pixels)
        sy = self.border + j * (self.width_per_env_pixels + self.spacing_pixels)
        self.height_field_raw[sx:sx + self.length_per_env_pixels, sy:sy + self.width_per_env_pixels] = terrain.height_field_raw
        ox = (i + 0.5) * self.env_length + i * self.spacing
        oy = (j + 0.5) * self.env_width + j * self.spacing
        hs = terrain.horizontal_scale
        x1, x2 = int((self.env_length / 2.0 - 1) / hs), int((self.env_length / 2.0 + 1) / hs)
        y1, y2 = int((self.env_width / 2.0 - 1) / hs), int((self.env_width / 2.0 + 1) / hs)
        self.env_origins[i, j] = [ox, oy, np.max(terrain.height_field_raw[x1:x2, y1:y2]) * terrain.vertical_scale]


_GENERATORS = {f.__name__: f for f in (random_uniform_terrain, pyramid_sloped_terrain, pyramid_stairs_terrain, discrete_obstacles_terrain,
                                       wave_terrain, stepping_stones_terrain, gap_terrain, pit_terrain)}
