#!/usr/bin/env python
"""An INDEPENDENT second restatement of FiniteStateMachine::FSMCallback — the per-drone state machine around replan() —
written straight from the reference's text WITHOUT reading oracle/, run on seeded sequences of (now, replan result,
isTrajSafe, goal reached); state, failure counter, traj_start_time_ and what is published after every tick are committed as
tests/golden/fsm_independent.json.  tests/test_fsm_independent.py holds the C++ oracle (`orc_fsm_tick`) and the tick driver's
tensorised rules (driver.fsm_plan_inputs / fsm_apply) to them.

Restated:  plan_manager/src/plan_manager.cpp:92-233, include/plan_manager/plan_manager.h:171-174 (checkTimeLapse: elapsed >
time, strict), :404-424 (publishEmptyTrajectory: start_time = traj_start_time_ AT THE CALL).  Inputs are never lost, the goal
is set and execution is triggered (the bench's situation).  Kept as the text has them:
  * NEW_PLAN plans only when more than 1.0 s have passed since traj_start_time_; it moves to EXEC_TRAJ on the MEMBER
    is_success_, which REPLAN never writes (`bool is_success_ = ...` there declares a local);
  * EXEC_TRAJ tests time lapse, safety and the goal one after the other — a reached goal wins;
  * REPLAN sets traj_start_time_ = now + replan_start_time before planning; the failure that exceeds replan_max_failures
    publishes the hover record FIRST (start time = now + replan_start_time) and only then sets traj_start_time_ = now - 1.0.
Run from the repo root:   python tests/golden/make_fsm_fixture.py
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPLAN_DURATION, REPLAN_START_TIME, MAX_FAILURES = 0.1, 0.02, 5     # sim_fake.yaml:7-10


class Fsm:
    def __init__(self, traj_start):
        self.status, self.traj_start, self.is_success, self.fails = "NEW_PLAN", traj_start, False, 0

    def lapse(self, now, d):
        return (now - self.traj_start) > d

    def tick(self, now, replan_ok, traj_safe, goal_reached):
        pub = None
        if self.status == "NEW_PLAN":
            if self.lapse(now, 1.0):
                self.traj_start = now
                self.is_success = bool(replan_ok)
                pub = ["new"] if self.is_success else ["hover", self.traj_start]
            if self.is_success:
                self.status = "EXEC_TRAJ"
        elif self.status == "EXEC_TRAJ":
            if self.lapse(now, REPLAN_DURATION):
                self.status = "REPLAN"
            if not traj_safe:
                self.status = "REPLAN"
            if goal_reached:
                self.status = "GOAL_REACHED"
        elif self.status == "REPLAN":
            self.traj_start = now + REPLAN_START_TIME
            if replan_ok:
                self.fails = 0
                pub = ["new"]
                self.status = "EXEC_TRAJ"
            else:
                self.fails += 1
                if self.fails > MAX_FAILURES:
                    self.status = "NEW_PLAN"
                    pub = ["hover", self.traj_start]
                    self.traj_start = now - 1.0
        return pub


def main():
    rng = np.random.default_rng(0xF5A)
    agents = []
    for a in range(24):
        p_ok = float(rng.uniform(0.05, 0.95))
        f = Fsm(98.0)
        ticks = []
        for k in range(120):
            now = 100.0 + 0.1 * k
            ok, safe, reached = bool(rng.random() < p_ok), bool(rng.random() < 0.9), bool(rng.random() < 0.004)
            pub = f.tick(now, ok, safe, reached)
            ticks.append([now, int(ok), int(safe), int(reached), f.status, f.fails, f.traj_start, pub])
            if f.status == "GOAL_REACHED":
                break
        agents.append(ticks)
    seen = {t[4] for ag in agents for t in ag} | {"hover" for ag in agents for t in ag if t[7] and t[7][0] == "hover"}
    print(len(agents), "agents,", sum(len(a) for a in agents), "ticks, seen", sorted(seen))
    path = os.path.join(HERE, "fsm_independent.json")
    json.dump({"what": "FSMCallback restated independently (tests/golden/make_fsm_fixture.py); a tick = [now, replan ok, traj safe, goal reached, status after, failures after, traj_start_time_ after, published: null | [new] | [hover, start time]]", "traj_start0": 98.0,
               "replan_duration": REPLAN_DURATION, "replan_start_time": REPLAN_START_TIME, "replan_max_failures": MAX_FAILURES,
               "agents": agents}, open(path, "w"))
    print("written", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
