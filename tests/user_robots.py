"""MJCF of two robots the reference does NOT ship — what a user of its plugin surface would bring (README.md:127 "define your own
robot", agent_model.py:12-41): a two-legged ant and a branching swimmer.  Neither has the shape of a built-in asset, so they
step on the generic tree kernel (csrc/generic_dyn.h).  Test data, written for these tests."""

BIPED_ANT = """
<mujoco model="biped_ant">
  <compiler angle="degree" coordinate="local" inertiafromgeom="true"/>
  <option integrator="RK4" timestep="0.02"/>
  <default>
    <joint armature="1" damping="1" limited="true"/>
    <geom conaffinity="0" condim="3" density="5.0" friction="1 0.5 0.5" margin="0.01" solimp="0.8 0.8 0.01" solref="0.02 1"/>
    <motor ctrllimited="true" ctrlrange="-20 20"/>
  </default>
  <worldbody>
    <geom name="floor" type="plane" size="40 40 40" conaffinity="1"/>
    <body name="torso" pos="0 0 0.6">
      <geom name="torso_geom" type="capsule" size="0.2" fromto="-0.25 0 0 0.25 0 0"/>
      <freejoint name="root"/>
      <body name="left_thigh" pos="0 0.2 0">
        <joint name="left_hip" type="hinge" axis="0 0 1" pos="0 0 0" range="-40 40"/>
        <geom name="left_thigh_geom" type="capsule" size="0.08" fromto="0 0 0 0.2 0.3 0"/>
        <body name="left_shin" pos="0.2 0.3 0">
          <joint name="left_knee" type="hinge" axis="-1 1 0" pos="0 0 0" range="30 70"/>
          <geom name="left_shin_geom" type="capsule" size="0.08" fromto="0 0 0 0.3 0.4 0"/>
        </body>
      </body>
      <body name="right_thigh" pos="0 -0.2 0">
        <joint name="right_hip" type="hinge" axis="0 0 1" pos="0 0 0" range="-40 40"/>
        <geom name="right_thigh_geom" type="capsule" size="0.08" fromto="0 0 0 0.2 -0.3 0"/>
        <body name="right_shin" pos="0.2 -0.3 0">
          <joint name="right_knee" type="hinge" axis="1 1 0" pos="0 0 0" range="30 70"/>
          <geom name="right_shin_geom" type="capsule" size="0.08" fromto="0 0 0 0.3 -0.4 0"/>
        </body>
      </body>
      <body name="tail" pos="-0.25 0 0">
        <joint name="tail_joint" type="hinge" axis="0 1 0" pos="0 0 0" range="-30 60"/>
        <geom name="tail_geom" type="capsule" size="0.06" fromto="0 0 0 -0.5 0 0"/>
        <geom name="tail_ball" type="sphere" size="0.1" pos="-0.55 0 0"/>
      </body>
    </body>
  </worldbody>
  <actuator>
    <motor joint="left_hip" gear="1"/>
    <motor joint="left_knee" gear="1"/>
    <motor joint="right_hip" gear="1"/>
    <motor joint="right_knee" gear="1"/>
    <motor joint="tail_joint" gear="0.5"/>
  </actuator>
</mujoco>
"""

# a swimmer whose mid link carries TWO tails (a Y): four links, a branching tree — not a chain the swimmer kernels could take
BRANCHING_SWIMMER = """
<mujoco model="y_swimmer">
  <compiler angle="degree" coordinate="local" inertiafromgeom="true"/>
  <option integrator="RK4" timestep="0.01" density="4000" viscosity="0.1" collision="predefined"/>
  <default>
    <geom conaffinity="1" condim="1" contype="1" density="1000"/>
    <joint armature="0.1"/>
  </default>
  <worldbody>
    <geom name="floor" type="plane" size="40 40 0.1" pos="0 0 -0.1" condim="3"/>
    <body name="torso" pos="0 0 0">
      <geom name="frontbody" type="capsule" size="0.1" fromto="1.5 0 0 0.5 0 0"/>
      <joint name="slider1" type="slide" axis="1 0 0" pos="0 0 0"/>
      <joint name="slider2" type="slide" axis="0 1 0" pos="0 0 0"/>
      <joint name="rot" type="hinge" axis="0 0 1" pos="0 0 0"/>
      <body name="mid" pos="0.5 0 0">
        <geom name="midbody" type="capsule" size="0.1" fromto="0 0 0 -1 0 0"/>
        <joint name="rot2" type="hinge" axis="0 0 1" pos="0 0 0" limited="true" range="-100 100"/>
        <body name="tail_a" pos="-1 0 0">
          <geom name="tail_a_geom" type="capsule" size="0.08" fromto="0 0 0 -0.7 0.4 0"/>
          <joint name="rot3" type="hinge" axis="0 0 1" pos="0 0 0" limited="true" range="-60 60"/>
        </body>
        <body name="tail_b" pos="-1 0 0">
          <geom name="tail_b_geom" type="capsule" size="0.08" fromto="0 0 0 -0.7 -0.4 0"/>
          <joint name="rot4" type="hinge" axis="0 0 1" pos="0 0 0" limited="true" range="-60 60"/>
        </body>
      </body>
    </body>
  </worldbody>
  <actuator>
    <motor joint="rot2" gear="150" ctrllimited="true" ctrlrange="-1 1"/>
    <motor joint="rot3" gear="100" ctrllimited="true" ctrlrange="-1 1"/>
    <motor joint="rot4" gear="100" ctrllimited="true" ctrlrange="-1 1"/>
  </actuator>
</mujoco>
"""


# a robot whose own limbs collide (MuJoCo's default: every geom pair but parent-child, contype = conaffinity = 1): two arms hinged side by side
# on a free-floating bar close onto each other — capsule-capsule contacts between sibling bodies, and a ball on a third arm for sphere-capsule
PINCER = """
<mujoco model="pincer">
  <compiler angle="degree" coordinate="local" inertiafromgeom="true"/>
  <option integrator="RK4" timestep="0.01"/>
  <default>
    <joint armature="0.5" damping="0.5" limited="true"/>
    <geom contype="1" conaffinity="1" condim="3" density="20.0" friction="1 0.5 0.5" margin="0.01" solimp="0.9 0.95 0.001" solref="0.02 1"/>
    <motor ctrllimited="true" ctrlrange="-10 10"/>
  </default>
  <worldbody>
    <geom name="floor" type="plane" size="40 40 40"/>
    <body name="bar" pos="0 0 0.3">
      <geom name="bar_geom" type="capsule" size="0.1" fromto="-0.3 0 0 0.3 0 0"/>
      <freejoint name="root"/>
      <body name="left_arm" pos="0.3 0.15 0">
        <joint name="left_hinge" type="hinge" axis="0 0 1" pos="0 0 0" range="-60 30"/>
        <geom name="left_arm_geom" type="capsule" size="0.06" fromto="0.12 0 0 0.6 0 0"/>
      </body>
      <body name="right_arm" pos="0.3 -0.15 0">
        <joint name="right_hinge" type="hinge" axis="0 0 1" pos="0 0 0" range="-30 60"/>
        <geom name="right_arm_geom" type="capsule" size="0.06" fromto="0.12 0 0 0.6 0 0"/>
      </body>
      <body name="top_arm" pos="0.3 0 0.12">
        <joint name="top_hinge" type="hinge" axis="0 1 0" pos="0 0 0" range="-20 60"/>
        <geom name="top_arm_geom" type="capsule" size="0.04" fromto="0.12 0 0 0.4 0 0"/>
        <geom name="top_ball" type="sphere" size="0.07" pos="0.5 0 0"/>
      </body>
    </body>
  </worldbody>
  <actuator>
    <motor joint="left_hinge" gear="1"/>
    <motor joint="right_hinge" gear="1"/>
    <motor joint="top_hinge" gear="1"/>
  </actuator>
</mujoco>
"""


# the biped with springs in its knees and tail (MJCF joint stiffness / springref): passive forces -k (q - springref)
SPRINGY_BIPED = BIPED_ANT.replace('name="left_knee" type="hinge"', 'name="left_knee" type="hinge" stiffness="6" springref="50"') \
                         .replace('name="right_knee" type="hinge"', 'name="right_knee" type="hinge" stiffness="6" springref="50"') \
                         .replace('name="tail_joint" type="hinge"', 'name="tail_joint" type="hinge" stiffness="3" springref="-10"') \
                         .replace('model="biped_ant"', 'model="springy_biped"')

# one arm on a vertical hinge fixed to the world, a torsion spring and no damping: a harmonic oscillator the oracle can be checked on
SPRING_ARM = """
<mujoco model="spring_arm">
  <compiler angle="degree" coordinate="local" inertiafromgeom="true"/>
  <option {integ} timestep="0.002"/>
  <default><geom conaffinity="0" contype="0" condim="3" density="500"/></default>
  <worldbody>
    <geom name="floor" type="plane" size="40 40 40" conaffinity="1"/>
    <body name="arm" pos="0 0 0.3">
      <joint name="pivot" type="hinge" axis="0 0 1" pos="0 0 0" armature="0.02" damping="{b}" stiffness="{k}" springref="{ref}"/>
      <geom name="arm_geom" type="capsule" size="0.03" fromto="0 0 0 0.4 0 0"/>
    </body>
  </worldbody>
  <actuator>{act}</actuator>
</mujoco>
"""
SPRING_ARM_MOTOR = '<motor joint="pivot" gear="1" ctrllimited="true" ctrlrange="-1 1"/>'


# the biped under MuJoCo's DEFAULT integrator (no integrator attribute = Euler: semi-implicit, implicit in the joint damping), a smaller step
EULER_BIPED = BIPED_ANT.replace('<option integrator="RK4" timestep="0.02"/>', '<option timestep="0.005"/>').replace('model="biped_ant"', 'model="euler_biped"')


def euler_class():
    from mujoco_maze_amd.agent_model import AgentModel

    class EulerBiped(AgentModel):
        ROBOT = "generic"
        FILE = EULER_BIPED
        MANUAL_COLLISION = False
        FRAME_SKIP = 8
        RESET_QVEL = "normal"

    return EulerBiped


# the biped with position servos on its knees and a velocity servo on the tail (MJCF <position kp>, <velocity kv>): PD-style control
SERVO_BIPED = BIPED_ANT.replace('<motor joint="left_knee" gear="1"/>', '<position joint="left_knee" kp="15" ctrllimited="true" ctrlrange="0.5 1.2"/>') \
                       .replace('<motor joint="right_knee" gear="1"/>', '<position joint="right_knee" kp="15" gear="1" ctrllimited="true" ctrlrange="0.5 1.2"/>') \
                       .replace('<motor joint="tail_joint" gear="0.5"/>', '<velocity joint="tail_joint" kv="2" gear="0.5" ctrllimited="true" ctrlrange="-3 3"/>') \
                       .replace('model="biped_ant"', 'model="servo_biped"')


def servo_class():
    from mujoco_maze_amd.agent_model import AgentModel

    class ServoBiped(AgentModel):
        ROBOT = "generic"
        FILE = SERVO_BIPED
        MANUAL_COLLISION = False
        FRAME_SKIP = 5
        RESET_QVEL = "normal"

    return ServoBiped


def springy_class():
    from mujoco_maze_amd.agent_model import AgentModel

    class SpringyBiped(AgentModel):
        ROBOT = "generic"
        FILE = SPRINGY_BIPED
        MANUAL_COLLISION = False
        FRAME_SKIP = 5
        RESET_QVEL = "normal"

    return SpringyBiped


def pincer_class():
    from mujoco_maze_amd.agent_model import AgentModel

    class Pincer(AgentModel):
        ROBOT = "generic"
        FILE = PINCER
        MANUAL_COLLISION = False
        FRAME_SKIP = 2
        RESET_QVEL = "normal"

    return Pincer


def robot_classes():
    from mujoco_maze_amd.agent_model import AgentModel

    class BipedAnt(AgentModel):
        ROBOT = "generic"
        FILE = BIPED_ANT
        MANUAL_COLLISION = False
        FRAME_SKIP = 5
        RESET_QVEL = "normal"

    class YSwimmer(AgentModel):
        ROBOT = "generic"
        FILE = BRANCHING_SWIMMER
        MANUAL_COLLISION = False
        FRAME_SKIP = 4
        RESET_QVEL = "uniform_sym"

    return BipedAnt, YSwimmer
