"""Environment factory with the reference's surface (/root/reference/environment/environment.py:11-102).

Only the maze is a device environment; lab / indoor actors are HOST-FED (hostfed_environment.py) by simulator objects
the caller supplies, because deepmind_lab / minos / gym are not in the image."""


class Environment(object):
    action_size = -1          # class-cached: first query wins (environment.py:13,46-47)
    LOG_DIR = None
    # stands in for minos.config.sim_config (indoor_environment.py:27-29): env_name -> {'objective_size': n}
    INDOOR_CONFIG = {}

    @staticmethod
    def register_indoor_config(env_name, objective_size):
        Environment.INDOOR_CONFIG[env_name] = {'objective_size': int(objective_size)}

    @staticmethod
    def create_environment(env_type, env_name, termination_time=50.0, env_args=None, thread_index=0):
        if env_type == 'maze':
            from . import maze_environment
            return maze_environment.MazeEnvironment()
        raise NotImplementedError("env_type %r needs an external simulator that is out of scope (SURVEY 8f)" % env_type)

    @staticmethod
    def get_action_size(env_type, env_name):
        if Environment.action_size >= 0:
            return Environment.action_size
        if env_type == 'maze':
            from . import maze_environment
            Environment.action_size = maze_environment.MazeEnvironment.get_action_size()
        elif env_type == 'lab':
            Environment.action_size = 6      # lab_environment.py:57-73
        elif env_type == 'indoor':
            Environment.action_size = 3      # indoor_environment.py:16-20
        else:
            raise NotImplementedError(env_type)
        return Environment.action_size

    @staticmethod
    def get_objective_size(env_type, env_name):
        if env_type == 'indoor':               # environment.py:68-72 -> indoor_environment.py:26-29
            return Environment.INDOOR_CONFIG.get(env_name, {}).get('objective_size', 0)
        return 0

    def __init__(self):
        pass

    def process(self, action):
        pass

    def reset(self):
        pass

    def stop(self):
        pass

    def is_all_scheduled_episodes_done(self):
        return False
