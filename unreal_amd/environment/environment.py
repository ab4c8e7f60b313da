"""Environment factory with the reference's surface (/root/reference/environment/environment.py:11-102).

Only the maze is a device environment this round; lab / indoor / gym need simulators that are not in
the image and raise."""


class Environment(object):
    action_size = -1          # class-cached: first query wins (environment.py:13,46-47)
    LOG_DIR = None

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
