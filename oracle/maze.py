"""Oracle: maze environment + pixel change (numpy).  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/environment/maze_environment.py:18-128 and
/root/reference/environment/environment.py:88-102.  Pinned by tests/golden/maze_*.npz.
"""
import numpy as np

MAP = ("--+---G"
       "--+-+++"
       "S-+---+"
       "--+++--"
       "--+-+--"
       "--+----"
       "-----++")          # maze_environment.py:18-25 (7x7, row-major, y down)
CELL = 12                  # 84 / 7
ACTION_DELTA = ((0, -1), (0, 1), (-1, 0), (1, 0))   # UP, DOWN, LEFT, RIGHT (:98-110)


def is_wall(x, y):
    return MAP[y * 7 + x] == '+'                     # :62-67


def find(ch):
    i = MAP.index(ch)
    return (i % 7, i // 7)


START = find('S')          # (0, 2)
GOAL = find('G')           # (6, 0)


def maze_image():
    """ch-0 wall blocks (maze_environment.py:30-48, 57-60)."""
    img = np.zeros((84, 84, 3), dtype=np.float64)
    for y in range(7):
        for x in range(7):
            if is_wall(x, y):
                img[CELL * y:CELL * y + CELL, CELL * x:CELL * x + CELL, 0] = 1.0
    return img


_MAZE_IMAGE = maze_image()


def render(x, y):
    """maze_environment.py:93-96: agent block painted into ch-1."""
    img = _MAZE_IMAGE.copy()
    img[CELL * y:CELL * y + CELL, CELL * x:CELL * x + CELL, 1] = 1.0
    return img


def move(x, y, action):
    """maze_environment.py:69-91 -> (new_x, new_y, hit)."""
    dx, dy = ACTION_DELTA[action]
    nx, ny = x + dx, y + dy
    cx = nx < 0 or nx > 6
    cy = ny < 0 or ny > 6
    nx = min(max(nx, 0), 6)
    ny = min(max(ny, 0), 6)
    hit_wall = False
    if is_wall(nx, ny):
        nx, ny = x, y
        hit_wall = True
    return nx, ny, (cx or cy or hit_wall)


def calc_pixel_change(state, last_state):
    """environment.py:88-102: |diff| on [2:-2,2:-2], channel mean, 4x4 block mean."""
    d = np.absolute(state[2:-2, 2:-2, :] - last_state[2:-2, 2:-2, :])
    m = np.mean(d, 2)
    s = m.shape
    return m.reshape(s[0] // 4, 4, s[1] // 4, 4).mean(-1).mean(1)


class OracleMaze(object):
    """Batch-1 environment with the reference's attribute surface."""

    action_size = 4

    def __init__(self):
        self.reset()

    def reset(self):                                  # :50-55
        self.x, self.y = START
        self.last_state = {'image': render(self.x, self.y)}
        self.last_action = 0
        self.last_reward = 0

    def process(self, action, flag=0):                # :98-128 (flag ignored: SURVEY H1)
        self.x, self.y, hit = move(self.x, self.y, int(action))
        image = render(self.x, self.y)
        terminal = (self.x, self.y) == GOAL
        reward = 1 if terminal else (-1 if hit else 0)
        pc = calc_pixel_change(image, self.last_state['image'])
        self.last_state = {'image': image}
        self.last_action = int(action)
        self.last_reward = reward
        return self.last_state, reward, terminal, pc

    def stop(self):
        pass
