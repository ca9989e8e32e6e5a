"""Random points of the CONFIGURATION space the reference exposes (gym_solo/core/configs.py:8-38: dt, motor torque
limit, start pose, gravity, damping, friction) plus the engine's own knobs (motor gains, contact ERP / margin, sweep
count): shared by the emulator (CPU) and the GPU parity tests - every field must reach the kernel the way it reaches
the oracle."""
import numpy as np


def random_config(seed):
  rng = np.random.default_rng(seed)
  return dict(
    dt=float(rng.choice([5e-4, 1e-3, 2e-3])),
    motor_torque_limit=float(rng.uniform(0.5, 4.0)),
    gravity=(float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), float(rng.uniform(-12, -6))),
    linear_damping=float(rng.uniform(0.0, 0.2)),
    angular_damping=float(rng.uniform(0.0, 0.2)),
    lateral_friction=float(rng.uniform(0.2, 1.2)),
    robot_start_pos=(float(rng.uniform(-0.1, 0.1)), float(rng.uniform(-0.1, 0.1)), float(rng.uniform(0.3, 0.6))),
    robot_start_orientation_euler=(float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-np.pi, np.pi))),
    motor_kp=float(rng.uniform(0.05, 0.3)),
    motor_kd=float(rng.uniform(0.5, 1.0)),
    contact_erp=float(rng.uniform(0.1, 0.4)),
    contact_margin=float(rng.uniform(0.002, 0.01)),
    solver_iterations=int(rng.choice([10, 30, 50])),
    settle_steps=int(rng.choice([40, 80])),
  )
