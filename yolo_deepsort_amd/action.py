"""Track-trajectory "action" heuristics that consume the tracker's int32 rows.

Host-side consumer of the hot path's output (reference action/action_Identify.py:4-47,
action/orbit.py:5-25, action/actions.py:24-150).  Pure Python, stateful and
wall-clock dependent like the reference; kept so that ``video_deepsort.py`` imports and runs.

Every motion action below answers one question about a track's recent foot points: did EVERY
consecutive step satisfy a predicate (the reference's flag loop reduces to exactly that: the
flag is raised by step 1 and any failing step clears it for good).
"""

from __future__ import annotations

import time
from collections import deque


class Orbit:
    """Recent foot points ((x1+x2)/2, y2) and their timestamps for one track id."""

    def __init__(self, max_age, track_id, class_id):
        self.max_age, self.track_id, self.class_id = max_age, track_id, class_id
        self.deque = deque(maxlen=max_age)
        self.timestamps = deque(maxlen=max_age)
        self.age = 0

    @staticmethod
    def _center_point(bbox):
        return bbox[0] + (bbox[2] - bbox[0]) / 2, bbox[3]

    def update(self, detection):
        self.age = 0
        self.deque.append(self._center_point(detection[:4]))
        self.timestamps.append(time.time())


class Action:
    def __init__(self, name):
        self.name = name

    def confirm(self, orbit):
        raise NotImplementedError


class _Stepwise(Action):
    """True when the orbit has >= 2 points of the right class and every step passes ``_step``."""

    def __init__(self, name, class_id):
        super().__init__(name)
        self.class_id = class_id

    def _step(self, prev, cur, dt):
        raise NotImplementedError

    def confirm(self, orbit):
        pts, ts = orbit.deque, orbit.timestamps
        if len(pts) < 2 or orbit.class_id != self.class_id:
            return False
        return all(self._step(pts[i - 1], pts[i], ts[i] - ts[i - 1]) for i in range(1, len(pts)))


class TakeOff(_Stepwise):
    def __init__(self, class_id, delta):
        super().__init__("takeoff", class_id)
        self.delta = delta

    def _step(self, prev, cur, dt):
        return prev[1] - cur[1] > self.delta[1] and abs(prev[0] - cur[0]) > self.delta[0]


class Landing(_Stepwise):
    def __init__(self, class_id, delta):
        super().__init__("landing", class_id)
        self.delta = delta

    def _step(self, prev, cur, dt):
        return cur[1] - prev[1] > self.delta[1] and abs(prev[0] - cur[0]) > self.delta[0]


class Glide(_Stepwise):
    def __init__(self, class_id, delta):
        super().__init__("glide", class_id)
        self.delta = delta

    def _step(self, prev, cur, dt):
        return abs(cur[1] - prev[1]) < self.delta[1] and abs(cur[0] - prev[0]) > self.delta[0]


class FastCrossing(_Stepwise):
    def __init__(self, class_id, speed):
        super().__init__("fast_crossing", class_id)
        self.speed = speed

    def _step(self, prev, cur, dt):
        return abs(cur[0] - prev[0]) / (dt * 1000) > self.speed


class BreakInto(Action):
    def __init__(self, class_id, timeout):
        super().__init__("break_into")
        self.class_id, self.timeout = class_id, timeout

    def confirm(self, orbit):
        return orbit.class_id == self.class_id and len(orbit.deque) > self.timeout


class ActionIdentify:
    def __init__(self, actions, max_age=30, max_size=4):
        self.cache = {}
        self.max_age, self.max_size, self.actions = max_age, max_size, actions

    def clone(self):
        return ActionIdentify(self.actions, self.max_age, self.max_size)

    def update(self, detections):
        if detections is None:
            return None
        seen = set()
        for det in detections:
            tid = det[4]
            seen.add(tid)
            if tid in self.cache:
                self.cache[tid].update(det)
            else:                                   # first sighting only registers the track
                self.cache[tid] = Orbit(self.max_size, tid, det[-1])
        for tid in [t for t in self.cache if t not in seen]:
            self.cache[tid].age += 1
            if self.cache[tid].age >= self.max_age:
                del self.cache[tid]
        return [(tid, o.class_id, a.name) for tid, o in self.cache.items() if o.age == 0
                for a in self.actions if a.confirm(o)]
