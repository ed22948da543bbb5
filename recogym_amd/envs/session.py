"""Organic session containers — reference: recogym/envs/session.py.

An OrganicSessions is a list of `{'t', 'u', 'z': 'pageview', 'v'}` dicts, one per organic
event, in event order; agents iterate it and index the dicts by key.
"""


class Session(list):
    def get_type(self):
        raise NotImplementedError

    def to_strings(self, user_id, session_id):
        kind = self.get_type()
        return [','.join([str(user_id), kind, str(session_id), row['z'], str(row['v'])])
                for row in self]


class OrganicSessions(Session):
    def next(self, context, product):
        self.append({'t': context.time(), 'u': context.user(), 'z': 'pageview', 'v': product})

    def get_type(self):
        return 'organic'

    def get_views(self):
        # (the reference unpacks the dict's KEYS here and therefore always returns [];
        #  nothing calls it — this returns the viewed products, the evident intent)
        return [row['v'] for row in self if row['z'] == 'pageview']
