"""Hyper-parameter flags with the reference's names and defaults.

Mirrors /root/reference/options.py:10-75 (`get_options(option_type)` -> flags object).  The
reference switches presets by copying a whole file over options.py; here `preset` selects them:
  "lab"     = /root/reference/options_lab.py  (segnet 0, pixel change on, n_step_TD 20, beta 1e-3)
              -- the preset BASELINE.json's configs are quoted on (default)
  "default" = /root/reference/options.py      (segnet 2, pixel change off, n_step_TD 50, beta 1e-4)
tf.app.flags is replaced by argparse (unknown argv entries are ignored, like absl does not)."""
import argparse
import os
import time


def _bool(v):
    return str(v).lower() in ("1", "true", "t", "yes", "y")


def get_options(option_type="training", preset="lab", argv=()):
    lab = preset == "lab"
    p = argparse.ArgumentParser(add_help=False)
    a = p.add_argument
    a("--env_type", default="lab"); a("--env_name", default="nav_maze_static_01")
    a("--use_lstm", type=_bool, default=True)
    a("--use_pixel_change", type=_bool, default=True if lab else False)
    a("--use_value_replay", type=_bool, default=True)
    a("--use_reward_prediction", type=_bool, default=True)
    a("--segnet_pretrain", type=_bool, default=False)
    a("--checkpoint_dir", default="lab_ckpt" if lab else "ckpt"); a("--checkpoint", default="")
    a("--segnet", type=int, default=0 if lab else 2)
    a("--segnet_config", default="config.json")
    a("--n_classes", type=int, default=9 if lab else 19)
    a("--termination_time_sec", type=float, default=50.0)
    a("--segnet_lambda", type=float, default=1.0)
    a("--dropout", type=float, default=0.3 if lab else 0.0)
    a("--parallel_size", type=int, default=8)
    a("--local_t_max", type=int, default=20)
    a("--n_step_TD", type=int, default=20 if lab else 50)
    a("--entropy_beta", type=float, default=0.001 if lab else 0.0001)
    if option_type == "training":
        a("--greedy_epsilon", type=float, default=0.99)
        a("--rmsp_alpha", type=float, default=0.99)
        a("--rmsp_epsilon", type=float, default=0.1)
        a("--log_dir", default=os.path.join("./logs", time.strftime("%Y_%m_%d_%H_%M_%S")))
        a("--initial_alpha_low", type=float, default=1e-4)
        a("--initial_alpha_high", type=float, default=5e-3)
        a("--initial_alpha_log_rate", type=float, default=0.5)
        a("--gamma", type=float, default=0.99)
        a("--gamma_pc", type=float, default=0.9)
        a("--pixel_change_lambda", type=float, default=0.05)
        a("--experience_history_size", type=int, default=2000)
        a("--max_time_step", type=int, default=int(13.2 * 10 ** 6))
        a("--save_interval_step", type=int, default=100 * 1000)
        a("--grad_norm_clip", type=float, default=40.0)
    if option_type == "display":
        a("--frame_save_dir", default="/tmp/unreal_frames")
        a("--recording", type=_bool, default=False)
        a("--frame_saving", type=_bool, default=False)
    if option_type in ("evaluate", "display"):
        a("--split", default="val")
        a("--episodes_per_scene", type=int, default=1)
        a("--log_action_trace", type=_bool, default=True)
    flags, _ = p.parse_known_args(list(argv))
    return flags
