"""Drop-in facades with the reference wrappers' surface (helpers/gridworld_gym_env.py etc.)."""
