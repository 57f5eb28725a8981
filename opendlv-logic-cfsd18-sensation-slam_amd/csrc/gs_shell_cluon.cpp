// gs_shell_cluon.cpp — main() of the drop-in microservice: the reference's process shell
// (src/opendlv-logic-cfsd18-sensation-slam.cpp:49-119) over the MI355X back-end.  Same command line, same OD4 session,
// the same seven data triggers; the Slam behind them is csrc/gs_slam.cpp + the HIP library.  Built by oracle/Makefile
// (target ref_shell) against the reference's libcluon header where it lies; see gs_shell_cluon.hpp.
#include <chrono>
#include <iostream>
#include <mutex>
#include <thread>

#include "gs_shell_cluon.hpp"

int32_t main(int32_t argc, char **argv) {
    auto commandlineArguments = cluon::getCommandlineArguments(argc, argv);
    if (commandlineArguments.size() < 10) {                        // the reference's own check and usage text (:52-57)
        gs_shell *none = nullptr;
        gs_shell_create(argc, argv, -2, &none);
        std::cerr << gs_last_error() << std::endl;
        return 1;
    }
    cluon::OD4Session od4{static_cast<uint16_t>(std::stoi(commandlineArguments["cid"]))};
    std::mutex mtx;                                                 // the shell is externally synchronised (the reference's m_optimizerMutex / m_mapMutex)
    ShellCluon shell(argc, argv, -1, [&od4](const gs_shell_msg &o) { ShellCluon::sendWith(od4, o); });
    if (shell.status() != GS_OK) { std::cerr << gs_last_error() << std::endl; return 1; }
    auto now_us = [] { return cluon::time::toMicroseconds(cluon::time::now()); };
    auto trigger = [&](cluon::data::Envelope &&envelope) { std::lock_guard<std::mutex> l(mtx); shell.onEnvelope(std::move(envelope), now_us()); };
    od4.dataTrigger(opendlv::proxy::GeodeticWgs84Reading::ID(), trigger);
    od4.dataTrigger(opendlv::proxy::GeodeticHeadingReading::ID(), trigger);
    od4.dataTrigger(opendlv::logic::sensation::Geolocation::ID(), trigger);
    od4.dataTrigger(opendlv::proxy::AngularVelocityReading::ID(), trigger);
    od4.dataTrigger(opendlv::logic::perception::ObjectDirection::ID(), trigger);
    od4.dataTrigger(opendlv::logic::perception::ObjectDistance::ID(), trigger);
    od4.dataTrigger(opendlv::logic::perception::ObjectType::ID(), trigger);
    // data driven like the reference; its per-frame gathering threads are this loop's poll (1 ms granularity)
    using namespace std::literals::chrono_literals;
    while (od4.isRunning()) {
        std::this_thread::sleep_for(1ms);
        std::lock_guard<std::mutex> l(mtx);
        if (shell.poll(now_us()) < 0) std::cerr << gs_last_error() << std::endl;
    }
    return 0;
}
