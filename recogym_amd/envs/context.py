"""Context of an observation: (time, user) — reference: recogym/envs/context.py."""


class Context:
    def time(self):
        raise NotImplementedError

    def user(self):
        raise NotImplementedError


class DefaultContext(Context):
    __slots__ = ('current_time', 'current_user_id')

    def __init__(self, current_time, current_user_id):
        self.current_time = current_time
        self.current_user_id = current_user_id

    def time(self):
        return self.current_time

    def user(self):
        return self.current_user_id
