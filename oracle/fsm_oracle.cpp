// fsm_oracle.cpp — CPU restatement of FiniteStateMachine::FSMCallback for ONE agent (row f2).
// TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Follows plan_manager/src/plan_manager.cpp:92-233 (the switch), checkTimeLapse
// (plan_manager/include/plan_manager/plan_manager.h:171-174) and publishEmptyTrajectory (:404-424), with the
// parameters of plan_manager/config/sim_fake.yaml:7-10.  What the node gets from ROS and the planner is an
// input here: `now` = ros::Time::now(), `replan_ok` = planner_->replan(...), `traj_safe` =
// planner_->isTrajSafe(colli_check_duration), `goal_reached` = isGoalReached(odom_pos_).  Inputs are never
// lost and execution is triggered (isInputLost() false, is_exec_triggered_ true: the batched driver has no
// RC trigger), so INIT / WAIT_TARGET are skipped.  Parity unpinned: the reference has no test of its FSM.
#include "oracle.h"

extern "C" void orc_fsm_init(OrcFsmState *s, double traj_start_time) {
  s->status               = ORC_FSM_NEW_PLAN;
  s->num_replan_failures  = 0;
  s->is_success           = 0;
  s->traj_start_time      = traj_start_time;
}

// One FSMCallback.  Returns the publication of this tick: 0 nothing, 1 publishTrajectory(),
// 2 publishEmptyTrajectory() with *hover_start_time = the traj_start_time_ it stamps (:412).
extern "C" int orc_fsm_tick(OrcFsmState *s, const OrcFsmConfig *cfg, double now, int replan_ok,
                            int traj_safe, int goal_reached, double *hover_start_time) {
  int pub = 0;
  switch (s->status) {
    case ORC_FSM_NEW_PLAN: {  // :109-134
      if (now - s->traj_start_time > 1.0) {  // checkTimeLapse(1.0): a new plan every second
        s->traj_start_time = now;
        s->is_success      = replan_ok ? 1 : 0;
        if (s->is_success) {
          pub = 1;
        } else {
          pub               = 2;
          *hover_start_time = s->traj_start_time;
        }
      }
      if (s->is_success) s->status = ORC_FSM_EXEC_TRAJ;  // is_exec_triggered_ && is_success_
      break;
    }
    case ORC_FSM_EXEC_TRAJ: {  // :137-163 (later conditions overwrite earlier ones)
      if (now - s->traj_start_time > cfg->replan_duration) s->status = ORC_FSM_REPLAN;
      if (!traj_safe) s->status = ORC_FSM_REPLAN;
      if (goal_reached) s->status = ORC_FSM_GOAL_REACHED;
      break;
    }
    case ORC_FSM_REPLAN: {  // :166-203
      s->traj_start_time = now + cfg->replan_start_time;
      if (replan_ok) {  // note: a LOCAL is_success_ shadows the member here (:179)
        s->num_replan_failures = 0;
        pub                    = 1;
        s->status              = ORC_FSM_EXEC_TRAJ;
      } else {
        s->num_replan_failures++;
        if (s->num_replan_failures > cfg->replan_max_failures) {
          s->status          = ORC_FSM_NEW_PLAN;
          pub                = 2;
          *hover_start_time  = s->traj_start_time;  // publishEmptyTrajectory reads it before the rewind
          s->traj_start_time = now - 1.0;           // "force new plan immediately" (:198)
        }
      }
      break;
    }
    default:  // GOAL_REACHED: the node shuts down (:224-231)
      break;
  }
  return pub;
}
