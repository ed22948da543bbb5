"""What `env.step` hands to an agent — reference: recogym/envs/observation.py."""


class Observation:
    __slots__ = ('current_context', 'current_sessions')

    def __init__(self, context, sessions):
        self.current_context = context
        self.current_sessions = sessions

    def context(self):
        return self.current_context

    def sessions(self):
        return self.current_sessions
